// Microbenchmark (MI355X, one wave on an otherwise idle CU): what a wave-wide broadcast of one double costs, per mechanism,
// on and off a dependency chain.  Input to the design of the banded substitutions of the batch kernel (batch_hip.hip ksolve).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/lane_bcast tools/lane_bcast_bench.hip && /tmp/lane_bcast
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__device__ __forceinline__ double readlane_f64(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}
template <int K>
__device__ __forceinline__ double rowbcast_f64(double v) {      // lane K of every row of 16 -> the row (DPP row_newbcast, gfx90a+)
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x150 + K, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x150 + K, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}

constexpr int kIters = 512;
// mode 0: dependent chain  readlane -> fma -> readlane ...   (the plain substitution recurrence)
// mode 1: 8 independent readlanes, then 8 fmas (two chains)   (per block of 8)
// mode 2: dependent chain  row_newbcast -> fma
// mode 3: fma chain alone
// mode 4: LDS: one ds_write_b64 (8 lanes), 4 broadcast ds_read_b128, 8 fmas, dependent block to block
// mode 5: 8 independent row_newbcasts, then 8 fmas
__global__ void k_bench(int mode, double *out, long long *cyc, const double *coef) {
  __shared__ __attribute__((aligned(16))) double sh[64];
  const int tid = threadIdx.x;
  double cur = 1.0 + 1e-3 * tid;
  double l[8];
#pragma unroll
  for (int q = 0; q < 8; q++) l[q] = coef[(tid + q) & 63];
  sh[tid] = cur;
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  if (mode == 0) {
    for (int it = 0; it < kIters; it++) {
#pragma unroll
      for (int q = 0; q < 8; q++) { const double v = readlane_f64(cur, (it * 8 + q) & 63); cur -= l[q] * v; }
    }
  } else if (mode == 1) {
    for (int it = 0; it < kIters; it++) {
      double c[8];
#pragma unroll
      for (int q = 0; q < 8; q++) c[q] = readlane_f64(cur, (it * 8 + q) & 63);
      double s0 = 0, s1 = 0;
#pragma unroll
      for (int q = 0; q < 8; q += 2) { s0 += l[q] * c[q]; s1 += l[q + 1] * c[q + 1]; }
      cur -= s0 + s1;
    }
  } else if (mode == 2) {
    for (int it = 0; it < kIters; it++) {
      cur -= l[0] * rowbcast_f64<0>(cur); cur -= l[1] * rowbcast_f64<1>(cur); cur -= l[2] * rowbcast_f64<2>(cur); cur -= l[3] * rowbcast_f64<3>(cur);
      cur -= l[4] * rowbcast_f64<4>(cur); cur -= l[5] * rowbcast_f64<5>(cur); cur -= l[6] * rowbcast_f64<6>(cur); cur -= l[7] * rowbcast_f64<7>(cur);
    }
  } else if (mode == 3) {
    for (int it = 0; it < kIters; it++) {
#pragma unroll
      for (int q = 0; q < 8; q++) cur -= l[q] * 1e-9;
    }
  } else if (mode == 4) {
    for (int it = 0; it < kIters; it++) {
      if (tid < 8) sh[tid] = cur;
      __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0)
      __builtin_amdgcn_wave_barrier();
      const double2 *p = reinterpret_cast<const double2 *>(sh);
      const double2 a = p[0], b = p[1], c = p[2], d = p[3];
      double s0 = l[0] * a.x + l[2] * b.x + l[4] * c.x + l[6] * d.x, s1 = l[1] * a.y + l[3] * b.y + l[5] * c.y + l[7] * d.y;
      cur -= s0 + s1;
      __builtin_amdgcn_wave_barrier();
    }
  } else if (mode == 5) {
    for (int it = 0; it < kIters; it++) {
      const double c0 = rowbcast_f64<0>(cur), c1 = rowbcast_f64<1>(cur), c2 = rowbcast_f64<2>(cur), c3 = rowbcast_f64<3>(cur);
      const double c4 = rowbcast_f64<4>(cur), c5 = rowbcast_f64<5>(cur), c6 = rowbcast_f64<6>(cur), c7 = rowbcast_f64<7>(cur);
      const double s0 = l[0] * c0 + l[2] * c2 + l[4] * c4 + l[6] * c6, s1 = l[1] * c1 + l[3] * c3 + l[5] * c5 + l[7] * c7;
      cur -= s0 + s1;
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  out[tid] = cur;
  if (tid == 0) cyc[mode] = t1 - t0;
}

int main() {
  double *out, *coef; long long *cyc;
  CK(hipMalloc(&out, 64 * sizeof(double))); CK(hipMalloc(&coef, 64 * sizeof(double))); CK(hipMalloc(&cyc, 8 * sizeof(long long)));
  double h[64]; for (int i = 0; i < 64; i++) h[i] = 1e-6 * (i + 1);
  CK(hipMemcpy(coef, h, sizeof(h), hipMemcpyHostToDevice));
  const char *names[6] = {"dependent chain: readlane_f64 -> fma (per pivot)", "8 independent readlane_f64 + 8 fma (per block of 8)",
                          "dependent chain: row_newbcast_f64 -> fma (per pivot)", "dependent fma chain alone (per fma)",
                          "LDS: write 8, 4 broadcast ds_read_b128, 8 fma (per block of 8)", "8 independent row_newbcast_f64 + 8 fma (per block of 8)"};
  const int per[6] = {8, 1, 8, 8, 1, 1};
  for (int rep = 0; rep < 2; rep++)
    for (int mode = 0; mode < 6; mode++) {
      hipLaunchKernelGGL(k_bench, dim3(1), dim3(64), 0, 0, mode, out, cyc, coef);
      CK(hipDeviceSynchronize());
      long long c; CK(hipMemcpy(&c, cyc + mode, sizeof(c), hipMemcpyDeviceToHost));
      if (rep) std::printf("%-66s %8.1f cycles\n", names[mode], (double)c / (kIters * per[mode]));
    }
  std::printf("(s_memtime ticks = shader cycles, MI355X_MICROARCH.md)\n");
  return 0;
}
