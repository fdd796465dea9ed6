// What a grid-wide barrier costs on this GPU against a kernel boundary: a cooperative launch of G workgroups x 256 threads that runs N
// cooperative_groups grid.sync() in a row (every workgroup also touches its own cache line between barriers, as a PCG phase would leave partials).
//   hipcc --offload-arch=gfx950 -O3 tools/grid_sync_bench.hip -o /tmp/grid_sync_bench && timeout 120 /tmp/grid_sync_bench
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <cstdio>
namespace cg = cooperative_groups;
__global__ __launch_bounds__(256) void k_sync(double *part, int n, double *out) {
  cg::grid_group grid = cg::this_grid();
  double acc = 0.0;
  for (int i = 0; i < n; i++) {
    if (threadIdx.x == 0) part[(size_t)blockIdx.x * 8 + (i & 1) * 8 * gridDim.x] = acc + i;
    grid.sync();
    if (threadIdx.x < 64) acc += part[(size_t)((blockIdx.x + threadIdx.x) % gridDim.x) * 8 + (i & 1) * 8 * gridDim.x];      // (reads what other workgroups wrote before the barrier)
  }
  if (threadIdx.x == 0) out[blockIdx.x] = acc;
}
__global__ __launch_bounds__(256) void k_empty(double *part, int i, double *out) {
  if (threadIdx.x == 0) part[(size_t)blockIdx.x * 8 + (i & 1) * 8 * gridDim.x] = i;
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main() {
  double *part, *out; CK(hipMalloc(&part, 8 * 2 * 8 * 2048)); CK(hipMalloc(&out, 8 * 2048)); CK(hipMemset(part, 0, 8 * 2 * 8 * 2048));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  int nb = 0; CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_sync, 256, 0));
  hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
  std::printf("%s: %d CUs, %d co-resident workgroups of 256 per CU\n", pr.name, pr.multiProcessorCount, nb);
  for (int G : {256, 512, 1024}) {
    if (G > nb * pr.multiProcessorCount) continue;
    int n = 2000; void *args[] = {&part, &n, &out};
    CK(hipLaunchCooperativeKernel(reinterpret_cast<void *>(k_sync), dim3(G), dim3(256), args, 0, 0)); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); CK(hipLaunchCooperativeKernel(reinterpret_cast<void *>(k_sync), dim3(G), dim3(256), args, 0, 0)); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    std::printf("grid of %4d workgroups: %.2f us per grid.sync() (+ one line written and 64 read per workgroup)\n", G, 1e3 * ms / n);
    CK(hipEventRecord(e0)); for (int i = 0; i < n; i++) hipLaunchKernelGGL(k_empty, dim3(G), dim3(256), 0, 0, part, i, out); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::printf("                          %.2f us per back-to-back launch of a kernel that writes one line per workgroup\n", 1e3 * ms / n);
  }
  return 0;
}
