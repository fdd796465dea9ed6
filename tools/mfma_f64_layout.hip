// Which lane / register of v_mfma_f64_16x16x4f64 holds which element?  A = e_i e_k' probes: for every (lane, operand) pattern print where a 1 lands.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_layout.hip -o /tmp/mfl && /tmp/mfl
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
__global__ void k(double *out) {
  const int l = threadIdx.x;
  // assumed operand layout: A[i = l % 16][k = l / 16], B[k = l / 16][j = l % 16]; A(i,k) = 100 i + k + 1, B(k,j) = (k == 0 ? 1 : 0) * (j + 1) * 0.001 ... use exact products
  const double a = (double)(l % 16) * 16.0 + (double)(l / 16);          // A(i,k) = 16 i + k
  const double b = (l / 16 == 2) ? (double)(l % 16 + 1) : 0.0;          // B(k,j) = (j + 1) for k = 2 only
  v4d c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);           // D(i,j) = A(i,2) B(2,j) = (16 i + 2)(j + 1)
  for (int r = 0; r < 4; r++) out[l * 4 + r] = c[r];
}
int main() {
  double *d, h[256]; hipMalloc(&d, sizeof(h)); hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d); hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int okA = 1, okB = 1;
  for (int l = 0; l < 64; l++) for (int r = 0; r < 4; r++) {
    const double v = h[l * 4 + r];
    // decode (i, j) from v = (16 i + 2)(j + 1): try both candidate layouts
    const int iA = 4 * (l / 16) + r, jA = l % 16, iB = (l / 16) + 4 * r, jB = l % 16;
    okA &= v == (16.0 * iA + 2) * (jA + 1); okB &= v == (16.0 * iB + 2) * (jB + 1);
  }
  std::printf("D layout i = 4 (lane / 16) + r, j = lane %% 16: %s;   i = lane / 16 + 4 r, j = lane %% 16: %s\n", okA ? "YES" : "no", okB ? "YES" : "no");
  for (int l : {0, 1, 16, 17, 33}) std::printf("lane %2d: %g %g %g %g\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
  return 0;
}
