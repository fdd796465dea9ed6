// Microbenchmark (MI355X): cost of one link in a chain of dependent, trivially-exiting kernels replayed from a hipGraph,
// as a function of grid shape and kernel-argument size.  This is the floor under every early-exit ("PCG already converged")
// launch of the engine and under every kernel boundary.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/launch_cost tools/launch_cost.hip && /tmp/launch_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

struct Small { const int *flag; double *out; };
struct Big { const int *flag; double *out; double pad[52]; };      // 432 bytes, like the engine's Dev

template <class A>
__global__ void k_noop(A a) { if (*a.flag) return; a.out[blockIdx.x * blockDim.x + threadIdx.x] = 1.0; }
template <class A>
__global__ void k_store(A a) { if (*a.flag == 2) return; if (threadIdx.x == 0) a.out[blockIdx.x] = 1.0; }   // one 8-byte store per workgroup

// the same early exit in front of a body that needs 32 KB of LDS / ~100 VGPRs (resources are allocated at dispatch even if unused)
template <class A>
__global__ __launch_bounds__(256) void k_noop_lds(A a) {
  __shared__ double buf[4096];
  if (*a.flag) return;
  buf[threadIdx.x] = a.out[threadIdx.x]; __syncthreads(); a.out[blockIdx.x * blockDim.x + threadIdx.x] = buf[(threadIdx.x * 7) & 4095];
}
template <class A>
__global__ __launch_bounds__(256) void k_noop_vgpr(A a) {
  if (*a.flag) return;
  double r[48];
#pragma unroll
  for (int i = 0; i < 48; i++) r[i] = a.out[threadIdx.x + 256 * i];
  double acc = 0;
#pragma unroll
  for (int i = 0; i < 48; i++) acc += r[i] * r[47 - i];
  a.out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
template <class A, class F>
float chain(F kern, int grid, int block, A arg, hipStream_t st, int links) {
  hipGraph_t g; hipGraphExec_t ex;
  hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
  for (int i = 0; i < links; i++) hipLaunchKernelGGL(kern, dim3(grid), dim3(block), 0, st, arg);
  hipStreamEndCapture(st, &g);
  hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipGraphLaunch(ex, st); hipStreamSynchronize(st);
  hipEventRecord(e0, st);
  for (int r = 0; r < 5; r++) hipGraphLaunch(ex, st);
  hipEventRecord(e1, st); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  hipGraphExecDestroy(ex); hipGraphDestroy(g); hipEventDestroy(e0); hipEventDestroy(e1);
  return ms * 1e3f / (5.f * links);
}

int main() {
  hipStream_t st; CK(hipStreamCreate(&st));
  int *flag; double *out;
  CK(hipMalloc(&flag, 4)); CK(hipMalloc(&out, sizeof(double) * 4096 * 1024));
  int one = 1; CK(hipMemcpy(flag, &one, 4, hipMemcpyHostToDevice));
  Small s{flag, out}; Big b{flag, out, {}};
  const int links = 400;
  std::printf("%-28s %10s %10s %10s %10s %10s\n", "grid x block", "noop-small", "noop-432B", "1store/wg", "noop+32KLDS", "noop+VGPRs");
  const int shapes[][2] = {{64, 256}, {256, 256}, {512, 256}, {1024, 256}, {2048, 256}, {256, 512}, {512, 512}, {256, 1024}, {1024, 64}, {1024, 128}, {2048, 128}};
  for (auto &sh : shapes) {
    float a = chain<Small>(k_noop<Small>, sh[0], sh[1], s, st, links);
    float c = chain<Big>(k_noop<Big>, sh[0], sh[1], b, st, links);
    float d = chain<Big>(k_store<Big>, sh[0], sh[1], b, st, links);
    float e = sh[1] == 256 ? chain<Big>(k_noop_lds<Big>, sh[0], sh[1], b, st, links) : 0.f;
    float f = sh[1] == 256 ? chain<Big>(k_noop_vgpr<Big>, sh[0], sh[1], b, st, links) : 0.f;
    std::printf("%5d x %-20d %10.2f %10.2f %10.2f %10.2f %10.2f   us per link\n", sh[0], sh[1], a, c, d, e, f);
  }
  return 0;
}
