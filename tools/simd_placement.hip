// Where do the waves of co-resident workgroups land (MI355X)?  The batch kernel (batch_hip.hip) runs the banded substitutions of a
// problem on ONE wave of its 256-thread workgroup, two workgroups per CU; v_readlane issues at ~14 cycles each and does not pipeline
// (tools/lane_bcast_bench.hip), so two substituting waves on the SAME SIMD would halve each other.  This probe launches the
// batch kernel's geometry (256 threads, 70 KB of LDS -> two workgroups per CU), records HW_ID of every wave, and times the dependent
// readlane chain on (a) wave 0 of every workgroup, (b) the wave sitting on SIMD (ticket & 3), ticket = a per-CU arrival counter.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/simd_placement tools/simd_placement.hip && /tmp/simd_placement
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__device__ __forceinline__ double readlane_f64(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}
constexpr int kPivots = 4096;

// mode 0: wave 0 runs the chain; mode 1: the wave whose SIMD_ID == ticket & 3 (fallback wave 0 if no wave sits there)
__global__ void __launch_bounds__(256, 2) k_probe(int mode, unsigned *hw, unsigned *xcc, long long *cyc, int *sel_out, int *tickets, double *out, const double *coef) {
  extern __shared__ double sh[];
  __shared__ int sel_s, simd_of[4];
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
  const unsigned id = __builtin_amdgcn_s_getreg(63492), xc = __builtin_amdgcn_s_getreg(63508);     // HW_REG_HW_ID, HW_REG_XCC_ID
  if (lane == 0) { hw[blockIdx.x * 4 + w] = id; xcc[blockIdx.x * 4 + w] = xc; simd_of[w] = (id >> 4) & 3; }
  sh[tid] = coef[tid & 63];
  __syncthreads();
  if (tid == 0) {
    int sel = 0;
    if (mode == 1) {
      const int cu = ((xc & 15) << 7) | (((id >> 13) & 7) << 4) | ((id >> 8) & 15);                // XCC, SE, CU
      const int t = atomicAdd(&tickets[cu], 1) & 3;
      for (int k = 0; k < 4; k++) if (simd_of[k] == t) { sel = k; break; }
    }
    sel_s = sel; sel_out[blockIdx.x] = sel;
  }
  __syncthreads();
  if (w == sel_s) {
    double cur = 1.0 + 1e-3 * lane;
    double l[8];
#pragma unroll
    for (int q = 0; q < 8; q++) l[q] = sh[(lane + q) & 63];
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < kPivots / 8; it++) {
#pragma unroll
      for (int q = 0; q < 8; q++) { const double v = readlane_f64(cur, (it * 8 + q) & 63); cur -= l[q] * v; }
    }
    const long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 64 + lane] = cur;
    if (lane == 0) cyc[blockIdx.x] = t1 - t0;
  }
  __syncthreads();
}

int main() {
  const int nb = 512;                                     // one round: two workgroups on each of 256 CUs
  unsigned *hw, *xcc; long long *cyc; int *sel, *tickets; double *out, *coef;
  CK(hipMalloc(&hw, nb * 4 * sizeof(unsigned))); CK(hipMalloc(&xcc, nb * 4 * sizeof(unsigned))); CK(hipMalloc(&cyc, nb * sizeof(long long)));
  CK(hipMalloc(&sel, nb * sizeof(int))); CK(hipMalloc(&tickets, 8192 * sizeof(int))); CK(hipMalloc(&out, nb * 64 * sizeof(double))); CK(hipMalloc(&coef, 64 * sizeof(double)));
  double h[64]; for (int i = 0; i < 64; i++) h[i] = 1e-6 * (i + 1);
  CK(hipMemcpy(coef, h, sizeof(h), hipMemcpyHostToDevice));
  const size_t lds = 70 * 1024;
  CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_probe), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  for (int mode = 0; mode < 2; mode++)
    for (int rep = 0; rep < 2; rep++) {
      CK(hipMemset(tickets, 0, 8192 * sizeof(int)));
      hipLaunchKernelGGL(k_probe, dim3(nb), dim3(256), lds, 0, mode, hw, xcc, cyc, sel, tickets, out, coef);
      CK(hipDeviceSynchronize());
      if (!rep) continue;
      std::vector<unsigned> H(nb * 4), X(nb * 4); std::vector<long long> C(nb); std::vector<int> S(nb);
      CK(hipMemcpy(H.data(), hw, H.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(X.data(), xcc, X.size() * 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(C.data(), cyc, C.size() * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(S.data(), sel, S.size() * 4, hipMemcpyDeviceToHost));
      std::map<int, std::vector<int>> cu_simds;           // CU -> SIMD of the substituting wave of each resident workgroup
      int distinct4 = 0; double csum = 0;
      for (int b = 0; b < nb; b++) {
        int mask = 0;
        for (int w = 0; w < 4; w++) mask |= 1 << ((H[b * 4 + w] >> 4) & 3);
        distinct4 += mask == 15;
        const unsigned id = H[b * 4 + S[b]], xc = X[b * 4 + S[b]];
        cu_simds[((xc & 15) << 7) | (((id >> 13) & 7) << 4) | ((id >> 8) & 15)].push_back((id >> 4) & 3);
        csum += (double)C[b];
      }
      int shared = 0, pairs = 0, cus = 0;
      for (auto &kv : cu_simds) { cus++; if (kv.second.size() == 2) { pairs++; shared += kv.second[0] == kv.second[1]; } }
      std::printf("mode %d (%s): %d workgroups on %d CUs; workgroups with their 4 waves on 4 distinct SIMDs: %d; CUs with two resident workgroups: %d, of which the two chains share a SIMD: %d; chain %.1f cycles per pivot (mean)\n",
                  mode, mode ? "wave on SIMD ticket&3" : "wave 0", nb, cus, distinct4, pairs, shared, csum / nb / kPivots);
      if (mode == 0) { std::printf("  first workgroups: wave->SIMD"); for (int b = 0; b < 6; b++) { std::printf("  [wg %d:", b); for (int w = 0; w < 4; w++) std::printf(" %u", (H[b * 4 + w] >> 4) & 3); std::printf("]"); } std::printf("\n"); }
    }
  return 0;
}
