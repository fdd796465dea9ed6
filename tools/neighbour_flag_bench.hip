// Prototype behind DESIGN.md section 8 "what comes next (0)": what ONE iteration costs when the launch boundary between PCG iterations is replaced by
// per-block flags inside a persistent kernel.  G workgroups x 256 threads, co-resident (cooperative launch, no grid.sync()); in iteration k block b
// waits until its neighbours b-D .. b+D have published iteration k-1, reads the 8 "vectors" on a window of W columns around its own C columns
// (written by those neighbours, most of them on the same XCD: the F1 block -> XCD mapping), passes four workgroup barriers (the LDS phases of the real
// kernel), writes 8 values for its own columns and one replica slice of W, and publishes k (release at agent scope: on gfx942 / gfx950 that is an
// L2 write-back, the acquire an invalidate -- the price of visibility across XCDs, and the thing this probe is here to measure).  Spins are bounded:
// a block that waits longer than ~0.2 s raises err and everybody leaves.
//   hipcc --offload-arch=gfx950 -O3 tools/neighbour_flag_bench.hip -o /tmp/nfb && timeout 120 /tmp/nfb
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int C = 100, W = 300, D = 4, NV = 8;
__global__ __launch_bounds__(256) void k_persist(int *flags, double *vec, size_t ns, int iters, int *err, int mode) {
  __shared__ double win[W * 2];
  const int G = gridDim.x, per = (G + 7) >> 3;
  const int b = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);        // XCD-contiguous ranges of blocks
  if (b >= G) return;
  const int tid = threadIdx.x;
  const int c0 = b * C, g0 = max(0, c0 - (W - C) / 2);
  for (int k = 1; k <= iters; k++) {
    if (mode >= 1 && tid <= 2 * D) {
      const int nb = b - D + tid;
      if (nb >= 0 && nb < G && nb != b) {
        long spins = 0;
        while (__hip_atomic_load(&flags[nb], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < k - 1) { if (++spins > 20000000 || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; } }
      }
    }
    if (mode >= 2) __atomic_thread_fence(__ATOMIC_ACQUIRE);      // (agent scope by default in HIP: see what the neighbours wrote)
    __syncthreads();
    if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
    const double *src = vec + (size_t)((k - 1) & 1) * NV * ns; double *dst = vec + (size_t)(k & 1) * NV * ns;
    double acc[2] = {0.0, 0.0};
    for (int u = 0; u < 2; u++) { const int e = tid + u * 256; if (e < W) { const int c = g0 + e; for (int v = 0; v < NV; v++) acc[u] += src[(size_t)v * ns + c]; win[e] = acc[u]; } }
    __syncthreads();
    double s = 0.0; for (int j = 0; j < 4; j++) s += win[(tid + j * 37) % W];
    __syncthreads(); win[W + tid % W] = s; __syncthreads();
    s += win[W + (tid * 3) % W]; __syncthreads();
    if (tid < C) for (int v = 0; v < NV - 1; v++) dst[(size_t)v * ns + c0 + tid] = 1e-3 * s + v;
    for (int u = 0; u < 2; u++) { const int e = tid + u * 256; if (e < W && (b % D == 0 || (g0 + e >= c0 && g0 + e < c0 + C))) dst[(size_t)(NV - 1) * ns + g0 + e] = 1e-3 * acc[u]; }
    if (mode >= 2) __atomic_thread_fence(__ATOMIC_RELEASE);
    __syncthreads();
    if (tid == 0) __hip_atomic_store(&flags[b], k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
// the same work as ONE launch per iteration (what the engine does): cold L1 / TLB at every launch.  chunk > 0: the 8 vectors interleaved in chunks of
// `chunk` columns (a window's 8 x 300 values within ~40 KB instead of in 8 places 800 KB apart)
__global__ __launch_bounds__(256) void k_once(double *vec, size_t ns, int k, int chunk) {
  __shared__ double win[W * 2];
  const int G = gridDim.x, per = (G + 7) >> 3;
  const int b = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
  if (b >= G) return;
  const int tid = threadIdx.x;
  const int c0 = b * C, g0 = max(0, c0 - (W - C) / 2);
  auto at = [&](int v, int c) -> size_t { return chunk ? (size_t)(c / chunk) * NV * chunk + (size_t)v * chunk + c % chunk : (size_t)v * ns + c; };
  const double *src = vec + (size_t)((k - 1) & 1) * NV * ns; double *dst = vec + (size_t)(k & 1) * NV * ns;
  double acc[2] = {0.0, 0.0};
  for (int u = 0; u < 2; u++) { const int e = tid + u * 256; if (e < W) { const int c = g0 + e; for (int v = 0; v < NV; v++) acc[u] += src[at(v, c)]; win[e] = acc[u]; } }
  __syncthreads();
  double s = 0.0; for (int j = 0; j < 4; j++) s += win[(tid + j * 37) % W];
  __syncthreads(); win[W + tid % W] = s; __syncthreads();
  s += win[W + (tid * 3) % W]; __syncthreads();
  if (tid < C) for (int v = 0; v < NV - 1; v++) dst[at(v, c0 + tid)] = 1e-3 * s + v;
  for (int u = 0; u < 2; u++) { const int e = tid + u * 256; if (e < W && (b % D == 0 || (g0 + e >= c0 && g0 + e < c0 + C))) dst[at(NV - 1, g0 + e)] = 1e-3 * acc[u]; }
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main() {
  const int G = 1024, iters = 2000; const size_t ns = (size_t)G * C + 512;
  int *flags, *err; double *vec; CK(hipMalloc(&flags, 4 * G)); CK(hipMalloc(&err, 4)); CK(hipMalloc(&vec, 8 * 2 * NV * ns));
  CK(hipMemset(vec, 0, 8 * 2 * NV * ns));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const char *what[3] = {"no synchronisation at all (the work alone; results meaningless)", "neighbour flags, no fences (ordering NOT guaranteed across XCDs)", "neighbour flags + acquire / release fences at agent scope"};
  for (int mode = 0; mode < 3; mode++) {
    for (int rep = 0; rep < 2; rep++) {
      CK(hipMemset(flags, 0, 4 * G)); CK(hipMemset(err, 0, 4));
      int it = iters, md = mode; size_t nsv = ns; void *args[] = {&flags, &vec, &nsv, &it, &err, &md};
      CK(hipEventRecord(e0)); CK(hipLaunchCooperativeKernel(reinterpret_cast<void *>(k_persist), dim3(G), dim3(256), args, 0, 0)); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1)); int herr = 0; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
      if (rep == 1) std::printf("%-70s %.2f us per iteration%s\n", what[mode], 1e3 * ms / iters, herr ? "  (a spin ran out: NOT co-resident or lost)" : "");
    }
  }
  for (int chunk : {0, 512}) {
    for (int rep = 0; rep < 2; rep++) {
      CK(hipEventRecord(e0));
      for (int k = 1; k <= iters; k++) hipLaunchKernelGGL(k_once, dim3(G), dim3(256), 0, 0, vec, ns, k, chunk);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep == 1) std::printf("one launch per iteration, vectors %-44s %.2f us per iteration\n", chunk ? "interleaved in chunks of 512 columns:" : "800 KB apart (the engine's layout):", 1e3 * ms / iters);
    }
  }
  return 0;
}
