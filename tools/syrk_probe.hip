// rocBLAS fp64 SYRK against GEMM for the large-rank Woodbury system (S = W W', W 10000 x 15000): is half the flops half the time?
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/syrk_probe tools/syrk_probe.hip -lrocblas && /tmp/syrk_probe
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>
#include <cstdio>
#include <vector>
int main() {
  const int r = 10000, k = 15000;
  double *W, *S;
  hipMalloc(&W, sizeof(double) * (size_t)r * k); hipMalloc(&S, sizeof(double) * (size_t)r * r);
  std::vector<double> h((size_t)r * k);
  for (size_t i = 0; i < h.size(); i++) h[i] = (double)((i * 2654435761u) & 0xffff) / 65536.0 - 0.5;
  hipMemcpy(W, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice);
  rocblas_handle hd; rocblas_create_handle(&hd);
  const double one = 1.0, zero = 0.0;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 3; mode++) {
    float best = 1e9f;
    for (int rep = 0; rep < 3; rep++) {
      hipEventRecord(e0);
      if (mode == 0) rocblas_dgemm(hd, rocblas_operation_transpose, rocblas_operation_none, r, r, k, &one, W, k, W, k, &zero, S, r);
      else if (mode == 1) rocblas_dsyrk(hd, rocblas_fill_lower, rocblas_operation_transpose, r, k, &one, W, k, &zero, S, r);
      else rocblas_dsyrk(hd, rocblas_fill_upper, rocblas_operation_transpose, r, k, &one, W, k, &zero, S, r);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    std::printf("%s: %.1f ms\n", mode == 0 ? "dgemm W'W" : (mode == 1 ? "dsyrk lower" : "dsyrk upper"), best);
  }
  return 0;
}
