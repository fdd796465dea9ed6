// engine_internal.hpp -- helpers shared by the translation units of the host driver (engine.cpp: policy, settings, scaling, driver loop, polish,
// solution; engine_setup.cpp: reordering, plans of the one-launch PCG form and of the Woodbury correction, setup; engine_api.cpp: updates,
// LinSysSolver slot, batch and small-problem paths, statistics).  Anonymous namespace: each unit has its own copy.
#pragma once
#include "engine.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <thread>
#include <limits>

namespace osqp_hip {

namespace {
constexpr double kRhoMin = 1e-6, kRhoMax = 1e6, kRhoTol = 1e-4;   // _osqp.py:25-28 (RHO_EQ_OVER_RHO_INEQ = 1e3 is applied in the set_rho kernel)
constexpr double kMinScaling = 1e-4, kMaxScaling = 1e4;                                 // _osqp.py:44-45
const double kNaN = std::numeric_limits<double>::quiet_NaN();

inline double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
inline double limit_scaling(double v) { return v < kMinScaling ? 1.0 : (v > kMaxScaling ? kMaxScaling : v); }   // _osqp.py:363-387
inline double clamp_rho(double r) { return std::min(std::max(r, kRhoMin), kRhoMax); }

template <class T>
T *dev_vec(Dev &d, size_t count) { return static_cast<T *>(be::alloc(d, std::max<size_t>(count, 1) * sizeof(T))); }

// Row blocks for the CSR-stream kernels: consecutive rows whose nnz sum to <= target (and <= kMaxRowsPerBlock rows);
// a row with more than kLongRow entries is a block of its own (reduced by the whole workgroup).
inline std::vector<int> build_row_blocks_target(const std::vector<int> &rowptr, int nrows, int target) {
  std::vector<int> rb; rb.push_back(0);
  int r = 0;
  while (r < nrows) {
    int len = rowptr[r + 1] - rowptr[r];
    if (len > kLongRow) { r++; rb.push_back(r); continue; }
    int start = r, acc = 0;
    while (r < nrows && r - start < kMaxRowsPerBlock) {
      int l2 = rowptr[r + 1] - rowptr[r];
      if (l2 > kLongRow || (acc + l2 > target && r > start)) break;
      acc += l2; r++;
    }
    rb.push_back(r);
  }
  return rb;
}
// Every kernel runs kGrid workgroups, so the number of row blocks is made a whole multiple k of kGrid with equal
// nnz per block (a 1172-block matrix on a 1024-workgroup grid would otherwise cost two full rounds).
// cap: most entries of a block (kChunk for the CSR-stream kernels; kF1Chunk when the one-launch PCG form is wanted: a workgroup then
// takes several blocks per launch on large problems)
inline std::vector<int> build_row_blocks(const std::vector<int> &rowptr, int nrows, int cap = kChunk) {
  const long nnz = nrows > 0 ? rowptr[nrows] : 0;
  long k = std::max<long>(1, (nnz + (long)kGrid * cap - 1) / ((long)kGrid * cap));
  for (;; k++) {
    int target = (int)std::max<long>(128, (nnz + kGrid * k - 1) / (kGrid * k));
    for (int attempt = 0; attempt < 40 && target <= cap; attempt++) {
      std::vector<int> rb = build_row_blocks_target(rowptr, nrows, target);
      if ((long)rb.size() - 1 <= kGrid * k) return rb;
      target = std::min<int>(cap + 1, target + std::max(1, target / 50));
    }
    if (k > 1024) return build_row_blocks_target(rowptr, nrows, cap);   // pathological (e.g. all rows long): accept
  }
}
// block descriptors {first row, end row, first nnz, end nnz}; long rows also get their run table (see DevCsr::runinfo).  Slices are the
// fixed kChunk steps the kernels take from the row's first entry
inline std::vector<int> block_descs(const std::vector<int> &rb, const std::vector<int> &rp, const std::vector<int> &cj, std::vector<int> &runs) {
  std::vector<int> d; d.reserve(4 * rb.size());
  runs.clear();
  for (size_t b = 0; b + 1 < rb.size(); b++) {
    const int r0 = rb[b], r1 = rb[b + 1], k0 = rp[r0], k1 = rp[r1];
    int end_row = r1;
    if (r1 - r0 == 1 && k1 - k0 > kLongRow) {
      end_row = -(1 + (int)runs.size());
      for (int base = k0; base < k1; base += kChunk) {
        const int end = std::min(k1, base + kChunk);
        bool run = true;
        for (int k = base + 1; k < end && run; k++) run = cj[k] == cj[k - 1] + 1;
        runs.push_back(run ? cj[base] : -1);
      }
    }
    d.push_back(r0); d.push_back(end_row); d.push_back(k0); d.push_back(k1);
  }
  return d;
}
}  // namespace

}  // namespace osqp_hip
