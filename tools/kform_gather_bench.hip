// Microbenchmark (MI355X): what bounds one launch of the K form (backend.h DevKf) on an unstructured matrix -- n = 100k rows, ~41 entries per row, columns
// uniformly random: 4.1 M random gathers of a column's record per launch next to a 49 MB matrix stream.
//   (a) gathers alone: 16 per thread, element size 8 / 16 / 32 bytes from a table of n elements (L2-resident: 0.8 / 1.6 / 3.2 MB)
//   (b) the ELL-16 product: every row padded to whole "slot rows" of 16 (column, value) pairs, a DPP row of 16 lanes per slot row, no LDS staging,
//       no barrier in the product phase; u = a - alpha (b + beta c) rebuilt from the gathered 32-byte record; row sums to LDS, own update per row.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/kgb tools/kform_gather_bench.hip && /tmp/kgb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); std::exit(1); } } while (0)

template <int EB>   // element bytes: 8, 16, 32
__global__ __launch_bounds__(256) void k_gather(const int *__restrict__ idx, const double *__restrict__ tab, double *__restrict__ out, int per) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  double acc = 0;
  for (int base = 0; base < per; base += 8) {
    int c[8];
#pragma unroll
    for (int u = 0; u < 8; u++) c[u] = idx[(size_t)(base + u) * gridDim.x * 256 + t];
    if (EB == 8) { double v[8];
#pragma unroll
      for (int u = 0; u < 8; u++) v[u] = tab[c[u]];
#pragma unroll
      for (int u = 0; u < 8; u++) acc += v[u]; }
    else if (EB == 16) { double2 v[8];
#pragma unroll
      for (int u = 0; u < 8; u++) v[u] = reinterpret_cast<const double2 *>(tab)[c[u]];
#pragma unroll
      for (int u = 0; u < 8; u++) acc += v[u].x + v[u].y; }
    else { double2 v[8], w[8];
#pragma unroll
      for (int u = 0; u < 8; u++) { v[u] = reinterpret_cast<const double2 *>(tab)[2 * (size_t)c[u]]; w[u] = reinterpret_cast<const double2 *>(tab)[2 * (size_t)c[u] + 1]; }
#pragma unroll
      for (int u = 0; u < 8; u++) acc += v[u].x + v[u].y + w[u].x + w[u].y; }
  }
  out[t] = acc;
}

// streaming read alone: 12 bytes per entry
__global__ __launch_bounds__(256) void k_stream(const int *__restrict__ col, const double *__restrict__ val, double *__restrict__ out, size_t total) {
  double acc = 0;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) acc += val[e] * (double)col[e];
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int CTRL> __device__ __forceinline__ double dppd(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double row16_sum(double v) {      // lane 15 of every DPP row of 16 holds the row's total
  v += dppd<0xb1>(v); v += dppd<0x4e>(v); v += dppd<0x114>(v); v += dppd<0x118>(v);
  return v;
}
// ELL-16: srp[r] = first slot row of row r; slot row s holds entries 16 s .. 16 s + 15 (padding: value 0, column = the row itself)
// Workgroup b owns rows [wr[b], wr[b + 1]); its 16 DPP rows take the slot rows of those rows round robin in batches of U.
template <int U, int MODE>   // MODE 0: 32-byte record + rebuild; 1: 8-byte gather of a ready vector
__global__ __launch_bounds__(256) void k_ell16(const int *__restrict__ wr, const int *__restrict__ srp, const int *__restrict__ col, const double *__restrict__ val,
                                               const unsigned char *__restrict__ last, const int *__restrict__ srow,
                                               const double *__restrict__ recr, double *__restrict__ recw, double *__restrict__ p, double *__restrict__ xs,
                                               const double *__restrict__ part, double *__restrict__ partw, double alpha, double beta, int fold) {
  __shared__ double wsum[512];
  __shared__ double red[16];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 15, grp = tid >> 4;
  const int r0 = wr[b], r1 = wr[b + 1];
  const int s0 = srp[r0], s1 = srp[r1];
  if (fold) {            // the scalar fold of the previous launch's partials: 3 x 1024 doubles per workgroup
    double a = 0, c = 0, d = 0;
    for (int k = tid; k < 1024; k += 256) { a += part[k]; c += part[1024 + k]; d = fmax(d, part[2048 + k]); }
    for (int o = 32; o; o >>= 1) { a += __shfl_xor(a, o); c += __shfl_xor(c, o); d = fmax(d, __shfl_xor(d, o)); }
    if ((tid & 63) == 0) { red[tid >> 6] = a; red[4 + (tid >> 6)] = c; red[8 + (tid >> 6)] = d; }
    __syncthreads();
    a = red[0] + red[1] + red[2] + red[3]; c = red[4] + red[5] + red[6] + red[7];
    if (a == 1.2345) alpha += c;      // (never true: keeps the fold alive)
  }
  double acc = 0.0;
  for (int sb = s0 + grp * U; sb < s1; sb += 16 * U) {
    int c[U]; double v[U];
#pragma unroll
    for (int u = 0; u < U; u++) { const int s = min(sb + u, s1 - 1); c[u] = col[(size_t)s * 16 + lane]; v[u] = val[(size_t)s * 16 + lane]; }
    double2 ga[U], gb[U]; double g1[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      if (MODE == 0) { ga[u] = reinterpret_cast<const double2 *>(recr)[2 * (size_t)c[u]]; gb[u] = reinterpret_cast<const double2 *>(recr)[2 * (size_t)c[u] + 1]; }
      else g1[u] = recr[c[u]];
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int s = sb + u;
      if (s < s1) {
        double un;
        if (MODE == 0) { const double sn = fma(beta, gb[u].y, gb[u].x), rn = fma(-alpha, sn, ga[u].y); un = ga[u].x * rn; } else un = g1[u];
        acc = fma(v[u], un, acc);
        const double tot = row16_sum(acc);
        const int fl = last[s];                 // 1: this slot row ends its row
        if (fl) { if (lane == 15) wsum[srow[s] - r0] = tot; acc = 0.0; }
      }
    }
  }
  __syncthreads();
  // own update of the workgroup's rows
  double g = 0, rnm = 0, dl = 0;
  for (int j = r0 + tid; j < r1; j += 256) {
    const double2 a = reinterpret_cast<const double2 *>(recr)[2 * (size_t)j], bb = reinterpret_cast<const double2 *>(recr)[2 * (size_t)j + 1];
    const double pp = p[j], px = xs[j], w = wsum[j - r0];
    const double sn = fma(beta, bb.y, bb.x), rn = fma(-alpha, sn, a.y), un = a.x * rn, pn = fma(beta, pp, a.x * a.y);
    xs[j] = fma(alpha, pn, px); p[j] = pn;
    reinterpret_cast<double2 *>(recw)[2 * (size_t)j] = make_double2(a.x, rn); reinterpret_cast<double2 *>(recw)[2 * (size_t)j + 1] = make_double2(w, sn);
    g += rn * un; rnm = fmax(rnm, fabs(rn)); dl += un * w;
  }
  for (int o = 32; o; o >>= 1) { g += __shfl_xor(g, o); dl += __shfl_xor(dl, o); rnm = fmax(rnm, __shfl_xor(rnm, o)); }
  __syncthreads();
  if ((tid & 63) == 0) { red[tid >> 6] = g; red[4 + (tid >> 6)] = dl; red[8 + (tid >> 6)] = rnm; }
  __syncthreads();
  if (tid == 0) { partw[b] = red[0] + red[1] + red[2] + red[3]; partw[1024 + b] = red[4] + red[5] + red[6] + red[7]; partw[2048 + b] = fmax(fmax(red[8], red[9]), fmax(red[10], red[11])); }
}

template <class F> float timeit(F f, hipStream_t st, int reps = 200) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 10; i++) f();
  hipEventRecord(e0, st);
  for (int i = 0; i < reps; i++) f();
  hipEventRecord(e1, st); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f / reps;
}

int main(int argc, char **argv) {
  const int n = argc > 1 ? std::atoi(argv[1]) : 100000;
  hipStream_t st; CK(hipStreamCreate(&st));
  std::mt19937 rng(1);
  // ---- (a) gathers alone
  { const int G = 1024, per = 16; const size_t tot = (size_t)G * 256 * per;
    std::vector<int> idx(tot); for (auto &v : idx) v = rng() % n;
    int *d_idx; double *tab, *out; CK(hipMalloc(&d_idx, tot * 4)); CK(hipMalloc(&tab, (size_t)n * 32)); CK(hipMalloc(&out, G * 256 * 8));
    CK(hipMemcpy(d_idx, idx.data(), tot * 4, hipMemcpyHostToDevice)); CK(hipMemset(tab, 0, (size_t)n * 32));
    std::printf("gathers alone, %zu per launch (n = %d):  8 B %.2f us   16 B %.2f us   32 B %.2f us\n", tot, n,
      timeit([&] { hipLaunchKernelGGL(k_gather<8>, dim3(G), dim3(256), 0, st, d_idx, tab, out, per); }, st),
      timeit([&] { hipLaunchKernelGGL(k_gather<16>, dim3(G), dim3(256), 0, st, d_idx, tab, out, per); }, st),
      timeit([&] { hipLaunchKernelGGL(k_gather<32>, dim3(G), dim3(256), 0, st, d_idx, tab, out, per); }, st));
    // sorted-within-thread-batch? no: the same with indices confined to a window of 4096 elements per workgroup (L1-friendly): the issue-rate floor
    for (size_t i = 0; i < tot; i++) { const size_t t = i % ((size_t)G * 256); const int wg = (int)(t / 256); idx[i] = (wg * 97 + (int)(rng() % 4096)) % n; }
    CK(hipMemcpy(d_idx, idx.data(), tot * 4, hipMemcpyHostToDevice));
    std::printf("  ... indices inside a 4096-element window per workgroup:  8 B %.2f us   32 B %.2f us\n",
      timeit([&] { hipLaunchKernelGGL(k_gather<8>, dim3(G), dim3(256), 0, st, d_idx, tab, out, per); }, st),
      timeit([&] { hipLaunchKernelGGL(k_gather<32>, dim3(G), dim3(256), 0, st, d_idx, tab, out, per); }, st));
    hipFree(d_idx); hipFree(tab); hipFree(out); }
  // ---- (b) ELL-16 product on a random symmetric-pattern-like matrix: row lengths 1 + 4 * Poisson(10)
  std::poisson_distribution<int> pois(10.0);
  std::vector<int> len(n); size_t nnz = 0; for (auto &l : len) { l = 1 + 4 * std::max(1, pois(rng)); nnz += l; }
  std::vector<int> srp(n + 1, 0); for (int r = 0; r < n; r++) srp[r + 1] = srp[r] + (len[r] + 15) / 16;
  const int ns = srp[n];
  std::vector<int> col((size_t)ns * 16), srow(ns); std::vector<double> val((size_t)ns * 16, 0.0); std::vector<unsigned char> last(ns, 0);
  for (int r = 0; r < n; r++) {
    for (int s = srp[r]; s < srp[r + 1]; s++) { srow[s] = r; for (int l = 0; l < 16; l++) col[(size_t)s * 16 + l] = r; }
    last[srp[r + 1] - 1] = 1;
    for (int k = 0; k < len[r]; k++) { col[(size_t)srp[r] * 16 + k] = rng() % n; val[(size_t)srp[r] * 16 + k] = 1e-3; }
  }
  // workgroups: 1024, equal slot rows
  const int G = 1024; std::vector<int> wr(G + 1, n); wr[0] = 0;
  { int r = 0; for (int b = 1; b < G; b++) { const long target = (long)ns * b / G; while (r < n && srp[r] < target) r++; wr[b] = r; } }
  std::printf("ELL-16: n = %d, nnz = %zu (%.1f per row), %d slot rows (%.1f %% padding), %.1f MB matrix\n", n, nnz, (double)nnz / n, ns, 100.0 * (16.0 * ns - nnz) / nnz, 12.0 * 16 * ns / 1e6);
  int *d_wr, *d_srp, *d_col, *d_srow; unsigned char *d_last; double *d_val, *rec, *p, *xs, *part, *out;
  CK(hipMalloc(&d_wr, (G + 1) * 4)); CK(hipMalloc(&d_srp, (n + 1) * 4)); CK(hipMalloc(&d_col, (size_t)ns * 64)); CK(hipMalloc(&d_val, (size_t)ns * 128)); CK(hipMalloc(&d_last, ns)); CK(hipMalloc(&d_srow, ns * 4));
  CK(hipMalloc(&rec, (size_t)n * 64)); CK(hipMalloc(&p, n * 8)); CK(hipMalloc(&xs, n * 8)); CK(hipMalloc(&part, 2 * 3 * 1024 * 8)); CK(hipMalloc(&out, 1024 * 256 * 8));
  CK(hipMemcpy(d_wr, wr.data(), (G + 1) * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_srp, srp.data(), (n + 1) * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_col, col.data(), (size_t)ns * 64, hipMemcpyHostToDevice)); CK(hipMemcpy(d_val, val.data(), (size_t)ns * 128, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_last, last.data(), ns, hipMemcpyHostToDevice)); CK(hipMemcpy(d_srow, srow.data(), ns * 4, hipMemcpyHostToDevice));
  CK(hipMemset(rec, 0, (size_t)n * 64)); CK(hipMemset(p, 0, n * 8)); CK(hipMemset(xs, 0, n * 8)); CK(hipMemset(part, 0, 2 * 3 * 1024 * 8));
  std::printf("matrix stream alone (12 B x %zu): %.2f us\n", (size_t)ns * 16, timeit([&] { hipLaunchKernelGGL(k_stream, dim3(1024), dim3(256), 0, st, d_col, d_val, out, (size_t)ns * 16); }, st));
  int k = 0;
  auto run = [&](auto kern, int fold) { return timeit([&] { const int cur = k & 1; k++; hipLaunchKernelGGL(kern, dim3(G), dim3(256), 0, st, d_wr, d_srp, d_col, d_val, d_last, d_srow, rec + (size_t)cur * 4 * n, rec + (size_t)(cur ^ 1) * 4 * n, p, xs, part + cur * 3072, part + (cur ^ 1) * 3072, 1e-3, 0.5, fold); }, st); };
  std::printf("ELL-16 product + own update, 32-byte records:  U=2 %.2f us   U=3 %.2f us   U=4 %.2f us   U=6 %.2f us;  with the fold: U=3 %.2f  U=4 %.2f us\n",
              run(k_ell16<2, 0>, 0), run(k_ell16<3, 0>, 0), run(k_ell16<4, 0>, 0), run(k_ell16<6, 0>, 0), run(k_ell16<3, 0>, 1), run(k_ell16<4, 0>, 1));
  std::printf("ELL-16 product with an 8-byte gather of a ready vector (what a second launch would see): U=4 %.2f us  U=6 %.2f us\n", run(k_ell16<4, 1>, 0), run(k_ell16<6, 1>, 0));
  return 0;
}
