// engine_setup.cpp -- Engine::setup and what it plans: the bandwidth-reducing reordering (compute_reorder / apply_reorder), the plan of the
// one-launch PCG form (plan_f1 / upload_f1), the plan of the Woodbury-corrected preconditioner and its two-launch direct mode (prepare_wb).
// See engine.hpp; formulas cite /root/reference/src/osqppurepy/_osqp.py.
#include "engine_internal.hpp"

namespace osqp_hip {

// ------------------------------------------------------------------------------------------------ reordering
// A QP whose band structure is hidden by the order in which its variables and constraints happen to be numbered takes the slow
// path (global gathers, two launches per PCG iteration) although a permutation would make it banded.  compute_reorder finds one:
//   1. breadth-first order of the COLUMNS through the bipartite graph of A (column -> its rows -> their columns) joined with P's
//      pattern, started from a pseudo-peripheral column (two sweeps), component by component (Cuthill-McKee levels);
//   2. three barycentre sweeps -- a row sits at the mean rank of its columns, a column moves to the mean position of its rows,
//      ranks are renewed by sorting -- which straighten the arbitrary order inside the BFS levels (measured on config 2 with shuffled
//      rows and columns: window of a 1000-entry row block 287 columns as generated, 331 after the BFS, 286 after two sweeps);
//   3. rows sorted by the middle of their (new) column range.
// O(nnz) per sweep + two sorts of n / m keys; runs only when the natural order does not admit the one-launch form.
bool Engine::compute_reorder(const std::vector<int> &Arp, const std::vector<int> &Arj, const std::vector<int> &Brp, const std::vector<int> &Bj) {
  std::vector<int> cstamp(n, 0), rstamp(m, 0), comp_done(n, 0), order, sweep, best;
  order.reserve(n);
  int stamp = 0;
  size_t widest = 0;                                 // widest BFS level seen (a lower bound of the bandwidth a level order can reach)
  auto bfs = [&](int start, std::vector<int> &out) {
    out.clear(); stamp++;
    out.push_back(start); cstamp[start] = stamp;
    size_t level_end = 1;
    for (size_t h = 0; h < out.size(); h++) {
      if (h == level_end) { widest = std::max(widest, out.size() - level_end); level_end = out.size(); }
      const int j = out[h];
      for (int k = A_.p[j]; k < A_.p[j + 1]; k++) {
        const int i = A_.i[k];
        if (rstamp[i] == stamp) continue;
        rstamp[i] = stamp;
        for (int e = Arp[i]; e < Arp[i + 1]; e++) { const int c = Arj[e]; if (cstamp[c] != stamp) { cstamp[c] = stamp; out.push_back(c); } }
      }
      for (int k = Brp[j]; k < Brp[j + 1] && Bj[k] < n; k++) { const int c = Bj[k]; if (cstamp[c] != stamp) { cstamp[c] = stamp; out.push_back(c); } }
    }
  };
  size_t ordered = 0;                                // columns of components whose level structure was worth ordering
  for (int s0 = 0; s0 < n; s0++) {
    if (comp_done[s0]) continue;
    widest = 0;                                      // (per component: one expander among the components must not hide a band in the others)
    bfs(s0, sweep);
    // An expander (columns drawn from everywhere) shows in the FIRST sweep: its levels explode, and no ordering of the levels can give row
    // blocks a window of kF1Win columns -- skip the two further sweeps (n = 100k unstructured: 97 ms of setup for nothing); the component
    // keeps its natural order
    if (widest > 8u * kF1Win) {
      std::sort(sweep.begin(), sweep.end());
      for (int c : sweep) { comp_done[c] = 1; order.push_back(c); }
      continue;
    }
    if (sweep.size() > 2) { bfs(sweep.back(), best); bfs(best.back(), sweep); }      // pseudo-peripheral start: the far end of the far end
    for (int c : sweep) { comp_done[c] = 1; order.push_back(c); }
    ordered += sweep.size();
  }
  if (2 * ordered < (size_t)n) return false;         // most of the problem is expander-like: no permutation will make the plan applicable
  std::vector<double> rank(n), prow(m), pcol(n);
  for (int k = 0; k < n; k++) rank[order[k]] = k;
  std::vector<int> idx(n);
  for (int it = 0; it < 3; it++) {
    for (int i = 0; i < m; i++) {
      const int cnt = Arp[i + 1] - Arp[i];
      double s = 0; for (int e = Arp[i]; e < Arp[i + 1]; e++) s += rank[Arj[e]];
      prow[i] = cnt ? s / cnt : 0.0;
    }
    for (int j = 0; j < n; j++) {
      const int cnt = A_.p[j + 1] - A_.p[j];
      double s = 0; for (int k = A_.p[j]; k < A_.p[j + 1]; k++) s += prow[A_.i[k]];
      pcol[j] = cnt ? s / cnt : rank[j];
    }
    for (int j = 0; j < n; j++) idx[j] = j;
    std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return pcol[a] < pcol[b] || (pcol[a] == pcol[b] && rank[a] < rank[b]); });
    for (int k = 0; k < n; k++) rank[idx[k]] = k;
  }
  pc_.assign(n, 0); ipc_.assign(n, 0);
  for (int j = 0; j < n; j++) { ipc_[j] = (int)rank[j]; pc_[(int)rank[j]] = j; }
  std::vector<long> key(m);
  for (int i = 0; i < m; i++) {
    int lo = INT32_MAX, hi = -1;
    for (int e = Arp[i]; e < Arp[i + 1]; e++) { lo = std::min(lo, ipc_[Arj[e]]); hi = std::max(hi, ipc_[Arj[e]]); }
    key[i] = hi >= 0 ? (long)lo + hi : 2L * n;                       // (empty rows last)
  }
  pr_.resize(m);
  for (int i = 0; i < m; i++) pr_[i] = i;
  std::stable_sort(pr_.begin(), pr_.end(), [&](int a, int b) { return key[a] < key[b]; });
  ipr_.assign(m, 0);
  for (int i = 0; i < m; i++) ipr_[pr_[i]] = i;
  return true;
}

// P_, A_, q0_, l0_, u0_ <- the permuted problem; PvalMap_ / AvalMap_ = where each of the caller's stored entries went.  Entries keep the
// caller's relative order inside a (row, column) pair (a CSC may repeat an entry), columns come out with ascending row indices.
void Engine::apply_reorder() {
  auto permute_csc = [&](HostCsc &M, const std::vector<int> &rmap, const std::vector<int> &cmap, bool upper, std::vector<int> &vmap) {
    const int nz = M.nnz(), nr = M.nr, nc = M.nc;
    std::vector<int> ri(nz), ci(nz);
    for (int j = 0; j < nc; j++)
      for (int k = M.p[j]; k < M.p[j + 1]; k++) {
        int r = rmap[M.i[k]], c = cmap[j];
        if (upper && r > c) std::swap(r, c);
        ri[k] = r; ci[k] = c;
      }
    // stable counting sort by row, then by column: sorted by (column, row), ties in the caller's order
    std::vector<int> byrow(nz), cnt(std::max(nr, nc) + 1, 0);
    for (int k = 0; k < nz; k++) cnt[ri[k] + 1]++;
    for (int r = 0; r < nr; r++) cnt[r + 1] += cnt[r];
    for (int k = 0; k < nz; k++) byrow[cnt[ri[k]]++] = k;
    HostCsc O; O.nr = nr; O.nc = nc; O.p.assign(nc + 1, 0); O.i.resize(nz); O.x.resize(nz);
    for (int k = 0; k < nz; k++) O.p[ci[k] + 1]++;
    for (int c = 0; c < nc; c++) O.p[c + 1] += O.p[c];
    std::vector<int> cur(O.p.begin(), O.p.end() - 1);
    vmap.assign(nz, 0);
    for (int t = 0; t < nz; t++) { const int k = byrow[t], pos = cur[ci[k]]++; O.i[pos] = ri[k]; O.x[pos] = M.x[k]; vmap[k] = pos; }
    M = std::move(O);
  };
  permute_csc(P_, ipc_, ipc_, true, PvalMap_);
  permute_csc(A_, ipr_, ipc_, false, AvalMap_);
  q0_ = to_internal_n(q0_.data()); l0_ = to_internal_m(l0_.data()); u0_ = to_internal_m(u0_.data());
  reordered_ = true;
}

void Engine::clear_reorder() {
  reordered_ = false;
  pc_.clear(); pr_.clear(); ipc_.clear(); ipr_.clear(); PvalMap_.clear(); AvalMap_.clear();
}

// ------------------------------------------------------------------------------------------------ F1 plan
// One launch per PCG iteration (backend.h DevF1): symbolic data, built once at setup from the row blocks of A.  The form applies when
// every row block of A has a column window of at most kF1Win columns (plus at most kF1MaxFar far columns outside it: per-block mixing,
// backend.h DevF1::mix) and at most kF1MaxRows rows, the windows of blocks g and g + D
// never overlap for some D <= kF1MaxD (banded / block-banded A -- as given, or after Engine::reorder has found the band), and the
// columns can be dealt out to the blocks as OWN columns -- consecutive ranges [cs[g], cs[g+1]) inside the block's window, at most
// kF1MaxOwn of them with at most kF1PChunk entries of P + sigma I.  Anything else keeps the two-kernel form.  OSQPHipPolicy::f1 = 0
// switches the plan off.  plan_f1 is host-only (no device state is touched: setup may try several row blockings / orderings).
bool Engine::plan_f1(const std::vector<int> &rb, const std::vector<int> &Arp, const std::vector<int> &Arj,
                     const std::vector<int> &Brp, const std::vector<int> &Bj, F1Plan &pl) {
  pl = F1Plan();
  if (!pol_.f1) return false;
  auto fail = [&](const char *why, int b) { if (pol_.setup_timing) std::fprintf(stderr, "[osqp_hip] F1 plan: %s (row block %d of %d)\n", why, b, (int)rb.size() - 1); return false; };
  const bool allow_mix = pol_.f1 != 2;                                        // (OSQPHipPolicy::f1 = 2: strict windows only, for A/B runs)
  const int nb = (int)rb.size() - 1;
  if (!be::device_assembly() || m == 0 || nb < kGrid / 4) return fail("too few row blocks", nb);      // (few blocks: most workgroups would idle in the vector update)
  std::vector<int> a0(nb), wl(nb), lo0(nb), hi0(nb), nfar(nb, 0), cs(nb + 1);
  std::vector<int> sc, dcols;
  int D = 0;
  // (up to three passes: 0 -- only blocks wider than kF1Win give columns away; 1 -- every block does (a block with ONE outlier 400 columns off fits
  //  kF1Win, but its window then overlaps those of many neighbours: more than kF1MaxD replicas); 2 -- the same with the tight choice, far-column
  //  cost 1 instead of 4: reaching out for near outliers keeps the far lists short but widens the windows)
  const int npass = allow_mix ? 3 : 1;
  for (int pass = 0; pass < npass; pass++) {
  bool pass_failed = false;                              // (a block this pass cannot place: the next pass gives more columns away -- only the last pass's failure is final)
  const int far_cost = pass == 2 ? 1 : 4;
  pl.mix = 0; std::fill(nfar.begin(), nfar.end(), 0);
  for (int b = 0; b < nb; b++) {
    const int r0 = rb[b], r1 = rb[b + 1], k0 = Arp[r0], k1 = Arp[r1];
    if (k1 == k0 || r1 - r0 > kF1MaxRows || k1 - k0 > kF1Chunk || (r1 - r0 == 1 && k1 - k0 > kLongRow)) return fail("block shape", b);
    int lo = INT32_MAX, hi = -1;
    for (int k = k0; k < k1; k++) { lo = std::min(lo, Arj[k]); hi = std::max(hi, Arj[k]); }
    if (hi - lo + 1 > kF1Win || pass >= 1) {
      // per-block mixing (backend.h DevF1::mix): the densest window that leaves room for the columns outside it as far columns
      if (!allow_mix) return fail("no window (strict)", b);
      sc.assign(Arj.begin() + k0, Arj.begin() + k1);
      std::sort(sc.begin(), sc.end());
      bool found = false;
      { // a far column costs about as much as kFarCost window columns (the same 4 + D loads, but scattered, behind a dependent index load, plus
        // its spill slot on both sides): the window [d_i, d_j] over the block's distinct columns that minimises  width + kFarCost * (columns outside)
        // = the maximum-sum run of  kFarCost + 1 - gap  over the gaps between neighbouring columns.  (Not "the window of kF1Win columns
        // holding the most entries": that stretches every window towards its nearest outliers, and wide windows cost replicas and loads.)
        const int kFarCost = far_cost;
        std::vector<int> &dc = dcols; dc.assign(sc.begin(), sc.end());
        dc.erase(std::unique(dc.begin(), dc.end()), dc.end());
        const int U = (int)dc.size();
        int bi = 0, bj = 0, ci = 0; long cur = 0, best = 0;
        for (int t = 1; t < U; t++) {
          cur += kFarCost + 1 - (dc[t] - dc[t - 1]);
          if (cur < 0) { cur = 0; ci = t; }
          else if (cur > best) { best = cur; bi = ci; bj = t; }
        }
        const int far = U - (bj - bi + 1);
        // (a run that holds less than half of the block's entries is no window -- columns spread at a constant stride have no positive run at all)
        const long inside = std::upper_bound(sc.begin(), sc.end(), dc[bj]) - std::lower_bound(sc.begin(), sc.end(), dc[bi]);
        if (far <= kF1MaxFar && 2 * inside >= (long)sc.size() && dc[bj] - dc[bi] + 1 <= (far ? kF1Win - kF1MaxFar : kF1Win)) { found = true; lo = dc[bi]; hi = dc[bj]; nfar[b] = far; }
      }
      if (!found && hi - lo + 1 <= kF1Win) found = true;                     // (a block that fits as it is)
      if (!found) {                                                            // ... or, failing that, the window of that width holding the most entries
        const int width = kF1Win - kF1MaxFar;
        int best = -1, bi = 0, bj = 0;
        for (int i = 0, j = 0; i < (int)sc.size(); i++) {                     // entries sc[i .. j) lie in [sc[i], sc[i] + width)
          while (j < (int)sc.size() && sc[j] - sc[i] < width) j++;
          if (j - i > best) { best = j - i; bi = i; bj = j; }
        }
        int far = 0;
        for (int i = 0; i < (int)sc.size(); i++) if ((i < bi || i >= bj) && (i == 0 || sc[i] != sc[i - 1])) far++;
        if (far <= kF1MaxFar) { found = true; lo = sc[bi]; hi = sc[bj - 1]; nfar[b] = far; }
      }
      if (!found) { if (pass + 1 < npass) { pass_failed = true; break; } return fail("too many far columns", b); }
      if (nfar[b]) pl.mix = 1;
    }
    lo0[b] = lo; hi0[b] = hi;
  }
  if (pass_failed) continue;
  // own columns: cs[g] follows the rows (rb[g] n / m: on a band of slope n / m these are the columns under the block) and is clamped into
  // what the neighbouring windows allow -- a column left of block g's window belongs to an earlier block, one right of block g - 1's
  // window to a later one; a column no window holds goes to the block in front of the gap
  cs[0] = 0;
  for (int g = 1; g < nb; g++) {
    const int ideal = (int)((long)rb[g] * n / m);
    const int lo = std::max(cs[g - 1], std::min(lo0[g], n)), up = hi0[g - 1] + 1;
    cs[g] = lo <= up ? std::min(std::max(ideal, lo), up) : lo;
    cs[g] = std::min(std::max(cs[g], cs[g - 1]), n);
  }
  cs[nb] = n;
  for (int b = 0; b < nb; b++) {
    // (the scatter window also covers the block's own columns: (P + sigma I) u of those joins the block's slice of A' t; a column
    //  without entries of the block's rows simply has an empty segment)
    int lo = lo0[b], hi = hi0[b];
    if (cs[b + 1] > cs[b]) { lo = std::min(lo, cs[b]); hi = std::max(hi, cs[b + 1] - 1); }
    if (hi - lo + 1 > (nfar[b] ? kF1Win - kF1MaxFar : kF1Win)) { if (pass + 1 < npass) { pass_failed = true; break; } return fail("window + own columns too wide", b); }
    a0[b] = lo; wl[b] = hi - lo + 1;
  }
  if (pass_failed) continue;
  D = 0;
  for (int t = 1; t <= kF1MaxD && !D; t++) {
    bool ok = true;
    for (int g = 0; g + t < nb && ok; g++) ok = a0[g] + wl[g] <= a0[g + t];
    if (ok) D = t;
  }
  if (D) break;
  }
  if (!D) return fail("windows of blocks g and g + 4 overlap", -1);
  // the compact CSR of P + sigma I (row j of B up to its first A' entry)
  std::vector<int> &prp = pl.prp; prp.assign(n + 1, 0);
  for (int j = 0; j < n; j++) { int c = 0; for (int k = Brp[j]; k < Brp[j + 1] && Bj[k] < n; k++) c++; prp[j + 1] = prp[j] + c; }
  const int pnnz = prp[n];
  std::vector<int> &pcol = pl.pcol, &psrc = pl.psrc; pcol.assign(std::max(pnnz, 1), 0); psrc.assign(std::max(pnnz, 1), 0);
  for (int j = 0; j < n; j++) for (int k = Brp[j], o = prp[j]; k < Brp[j + 1] && Bj[k] < n; k++, o++) { pcol[o] = Bj[k]; psrc[o] = k; }
  std::vector<int> &blk = pl.blk; blk.assign(16 * (size_t)nb, 0);
  std::vector<unsigned int> &ent = pl.ent; ent.assign(Arj.size(), 0u);
  std::vector<unsigned short> &cptr = pl.cptr; cptr.clear();
  if (pl.mix) { pl.fcol.assign((size_t)nb * kF1MaxFar, -1); pl.fq.assign((size_t)nb * kF1MaxFar, 0); }
  size_t nsp = 0;
  std::vector<int> order, tpos, fc, slot;
  for (int b = 0; b < nb; b++) {
    if (cs[b + 1] - cs[b] > kF1MaxOwn || prp[cs[b + 1]] - prp[cs[b]] > kF1PChunk) return fail("own columns", b);
    const int r0 = rb[b], r1 = rb[b + 1], k0 = Arp[r0], k1 = Arp[r1], cnt = k1 - k0;
    int *w = &blk[16 * (size_t)b];
    w[0] = r0; w[1] = r1; w[2] = k0; w[3] = k1;
    w[4] = b < D ? 0 : a0[b]; w[5] = b + D < nb ? a0[b + D] : n; w[6] = cs[b]; w[7] = cs[b + 1];
    w[8] = (int)cptr.size(); w[9] = prp[cs[b]]; w[10] = prp[cs[b + 1]]; w[11] = 0;      // (w[11]: far columns, below)
    // far columns: whatever the final scatter window [a0, a0 + wl) does not hold (the own columns may have widened it over some)
    fc.clear();
    if (nfar[b]) {
      for (int k = k0; k < k1; k++) if (Arj[k] < a0[b] || Arj[k] >= a0[b] + wl[b]) fc.push_back(Arj[k]);
      std::sort(fc.begin(), fc.end()); fc.erase(std::unique(fc.begin(), fc.end()), fc.end());
    }
    const int nfc = (int)fc.size();
    w[11] = nfc;
    // gather window: the columns of the block's rows of A, together with those of its own rows of P + sigma I when that widens the
    // window by at most a quarter (every window column costs 4 + D vector loads; a P entry outside the window costs as many, once)
    int g0 = a0[b], g1 = a0[b] + wl[b];
    for (int k = prp[cs[b]]; k < prp[cs[b + 1]]; k++) { g0 = std::min(g0, pcol[k]); g1 = std::max(g1, pcol[k] + 1); }
    if (g1 - g0 > (nfc ? kF1Win - kF1MaxFar : kF1Win) || 4 * (g1 - g0) > 5 * wl[b]) { g0 = a0[b]; g1 = a0[b] + wl[b]; }
    w[12] = g0; w[13] = g1 - g0; w[14] = a0[b]; w[15] = wl[b];
    // per entry: the slot of its column in the gather list (window columns first, then the far columns) and in the column-ordered pass
    slot.resize(cnt);
    for (int e = 0; e < cnt; e++) {
      const int c = Arj[k0 + e];
      if (c >= a0[b] && c < a0[b] + wl[b]) slot[e] = c - a0[b];
      else slot[e] = wl[b] + (int)(std::lower_bound(fc.begin(), fc.end(), c) - fc.begin());
    }
    // column-major order of the block's entries: stable by slot (rows ascending within a column)
    order.resize(cnt);
    for (int e = 0; e < cnt; e++) order[e] = e;
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return slot[x] < slot[y]; });
    tpos.resize(cnt);
    for (int t = 0; t < cnt; t++) tpos[order[t]] = t;
    for (int r = r0; r < r1; r++)
      for (int k = Arp[r]; k < Arp[r + 1]; k++) {
        const int sl = slot[k - k0];
        const unsigned gs = sl < wl[b] ? (unsigned)(Arj[k] - g0) : (unsigned)(kF1Win - kF1MaxFar + sl - wl[b]);      // (far columns: the last kF1MaxFar gather slots)
        ent[k] = gs | ((unsigned)(r - r0) << 9) | ((unsigned)tpos[k - k0] << 18);
      }
    const size_t base = cptr.size();
    cptr.resize(base + wl[b] + nfc + 1, 0);
    for (int e = 0; e < cnt; e++) cptr[base + slot[e] + 1]++;
    for (int c = 0; c < wl[b] + nfc; c++) cptr[base + c + 1] = (unsigned short)(cptr[base + c + 1] + cptr[base + c]);
    if (pl.mix) { std::copy(fc.begin(), fc.end(), pl.fcol.begin() + (size_t)b * kF1MaxFar); nsp += (size_t)nfc; }
  }
  if (pol_.f1 == 3 && !pl.mix) { pl.mix = 1; pl.fcol.assign((size_t)nb * kF1MaxFar, -1); pl.fq.assign((size_t)nb * kF1MaxFar, 0); }      // (experiment: the mixing kernels on a matrix without far columns)
  if (pl.mix && nsp >= (size_t)1 << 25) return fail("too many spill slots", -1);
  if (pl.mix) {
    // spill slots ordered by (column, block): blocks ascend within fcol, so a stable counting sort by column gives the order
    pl.nsp = nsp;
    pl.sp_ptr.assign((size_t)n + 1, 0);
    for (int c : pl.fcol) if (c >= 0) pl.sp_ptr[c + 1]++;
    for (int j = 0; j < n; j++) pl.sp_ptr[j + 1] += pl.sp_ptr[j];
    std::vector<int> cur(pl.sp_ptr.begin(), pl.sp_ptr.end() - 1);
    for (size_t i = 0; i < pl.fcol.size(); i++) if (pl.fcol[i] >= 0) pl.fq[i] = cur[pl.fcol[i]]++;
    for (int j = 0; j < n; j++) if (pl.sp_ptr[j + 1] - pl.sp_ptr[j] > 63) return fail("a column is far for more than 63 row blocks", -1);
  }
  pl.D = D; pl.pnnz = pnnz; pl.ok = true;
  if (pol_.setup_timing) {
    double swl = 0, sgl = 0, sown = 0; int mwl = 0;
    for (int b = 0; b < nb; b++) { swl += wl[b]; sgl += blk[16 * (size_t)b + 13]; sown += cs[b + 1] - cs[b]; mwl = std::max(mwl, wl[b]); }
    std::fprintf(stderr, "[osqp_hip] F1 plan: %d row blocks, D = %d, scatter window %.1f columns on average (max %d), gather window %.1f, own columns %.1f, far columns %.2f per block\n",
                 nb, D, swl / nb, mwl, sgl / nb, sown / nb, (double)pl.nsp / nb);
  }
  return true;
}

void Engine::upload_f1(const F1Plan &pl) {
  d_.f1 = DevF1();
  if (!pl.ok) return;
  auto up_i = [&](const std::vector<int> &h) { int *p = dev_vec<int>(d_, h.size()); if (!h.empty()) be::h2d(d_, p, h.data(), sizeof(int) * h.size()); return p; };
  DevF1 &f = d_.f1;
  f.D = pl.D; f.pnnz = pl.pnnz;
  f.blk = up_i(pl.blk); f.prp = up_i(pl.prp); f.pcol = up_i(pl.pcol); f.psrc = up_i(pl.psrc);
  { // the blocks' matrix streams (backend.h DevF1::stream): the index words now, the values by be::f1_refresh once A.val is assembled
    const size_t nb = pl.blk.size() / 16;
    std::vector<unsigned char> hs(nb * (size_t)kF1StreamBytes, 0);
    for (size_t b = 0; b < nb; b++) {
      const int k0 = pl.blk[16 * b + 2], cnt = pl.blk[16 * b + 3] - k0;
      std::memcpy(hs.data() + b * (size_t)kF1StreamBytes + sizeof(double) * kF1Chunk, pl.ent.data() + k0, sizeof(unsigned int) * (size_t)cnt);
    }
    f.stream = dev_vec<unsigned char>(d_, hs.size()); be::h2d(d_, f.stream, hs.data(), hs.size());
  }
  f.cptr = dev_vec<unsigned short>(d_, pl.cptr.size()); be::h2d(d_, f.cptr, pl.cptr.data(), sizeof(unsigned short) * pl.cptr.size());
  f.pval = dev_vec<double>(d_, pl.pnnz);
  f.ns = ((size_t)n + 31) / 32 * 32;                       // 256-byte aligned vectors
  f.va = dev_vec<double>(d_, (7 + 3 * (size_t)pl.D) * f.ns);
  if (pl.mix) {
    std::vector<int> spk((size_t)n), fc2(2 * pl.fcol.size(), 0);
    for (int j = 0; j < n; j++) spk[j] = (pl.sp_ptr[j] << 6) | (pl.sp_ptr[j + 1] - pl.sp_ptr[j]);
    for (size_t i = 0; i < pl.fcol.size(); i++) if (pl.fcol[i] >= 0) { fc2[2 * i] = pl.fcol[i]; fc2[2 * i + 1] = spk[pl.fcol[i]]; }
    f.mix = 1; f.fcol = up_i(fc2); f.fq = up_i(pl.fq); f.sp_ptr = up_i(pl.sp_ptr); f.spk = up_i(spk);
    f.nsp = pl.nsp; f.spill = dev_vec<double>(d_, 3 * (f.nsp + 2));
  }
  { // write-through stores (DevF1::wt) while what a launch touches stays in the 256 MiB Infinity Cache: stream + vector arena + the m-vectors KA writes
    const double mb = ((double)(pl.blk.size() / 16) * kF1StreamBytes + 8.0 * (7 + 3 * (double)pl.D) * (double)f.ns + 8.0 * 10 * (double)m) / (1024.0 * 1024.0);
    const char *e = std::getenv("OSQP_HIP_F1_WT");
    // (measured, F launch in us with the results written through / with plain stores and a non-temporal matrix stream -- the kernels' two states:
    //  n = 250k (110 MB by this estimate) 26.8 / 28.7;  400k (175 MB) 34.9 / 37.8;  550k (241 MB) 47.8 / 51.3;  750k (328 MB) 62.8 / 66.3;  1M (438 MB) 98.0 / 82.1.
    //  Plain stores WITHOUT the non-temporal stream would be the best of three at 550k - 750k (47.1 / 59.9 us), but a third instantiation of every F1 kernel
    //  cost the n = 100k solve 0.25 ms in the same-box A/B -- not kept)
    f.wt = e ? (e[0] != '0') : (mb <= 380.0);
  }
  f.on = 1;
}

// ------------------------------------------------------------------------------------------------ K form
// backend.h DevKf: the pattern of K = P + sigma I + A' diag(rho) A and, per entry, the list of its terms -- built once, row by row (Gustavson):
// row j of K collects the (P + sigma I) entries of row j of B and, for every entry (i, j) of column j of A, the products A_ij A_ic over row i.
// Terms of an entry are kept in that order (B entries first, then rows i ascending, row entries in CSR order): k_kf_values sums them in list
// order, so K.val is a deterministic function of (rho, A.val, B.val).  A valid CSC may repeat an entry: every stored pair makes its own term.
void Engine::prepare_kf(const std::vector<int> &Arp, const std::vector<int> &Arj, const std::vector<int> &Brp, const std::vector<int> &Bj) {
  d_.kf = DevKf();
  if (!pol_.kform || !be::kf_supported() || !be::device_assembly() || m == 0) return;
  const double t0 = now_s();
  const long nzA = Arp[m];
  long nprod = 0;
  for (int i = 0; i < m; i++) { const long len = Arp[i + 1] - Arp[i]; nprod += len * len; if (nprod > (long)kKfMaxFill * nzA + n) break; }
  long npt = 0;
  for (int j = 0; j < n; j++) for (int k = Brp[j]; k < Brp[j + 1] && Bj[k] < n; k++) npt++;
  if (nprod > (long)kKfMaxFill * nzA + n || nprod + npt >= (1L << 30)) {
    if (pol_.setup_timing) std::fprintf(stderr, "[osqp_hip] K form: fill too large (sum of squared row lengths > %d nnz(A))\n", kKfMaxFill);
    return;
  }
  std::vector<int> Krp(n + 1, 0), Kj, tptr, trow, ta, tb;
  Kj.reserve((size_t)(nprod + npt)); trow.reserve((size_t)(nprod + npt)); ta.reserve((size_t)(nprod + npt)); tb.reserve((size_t)(nprod + npt));
  tptr.reserve((size_t)(nprod + npt) + 1);
  std::vector<int> mark(n, -1), slot(n, 0), cols, cnt, order, newslot, start;
  struct Term { int s, row, a, b; };
  std::vector<Term> terms;
  for (int j = 0; j < n; j++) {
    cols.clear(); cnt.clear(); terms.clear();
    auto add = [&](int c, int row, int a, int b) {
      if (mark[c] != j) { mark[c] = j; slot[c] = (int)cols.size(); cols.push_back(c); cnt.push_back(0); }
      cnt[slot[c]]++; terms.push_back(Term{slot[c], row, a, b});
    };
    for (int k = Brp[j]; k < Brp[j + 1] && Bj[k] < n; k++) add(Bj[k], -1, k, 0);
    for (int k = A_.p[j]; k < A_.p[j + 1]; k++) {
      const int i = A_.i[k], pa = AmapA_[k];
      for (int e = Arp[i]; e < Arp[i + 1]; e++) add(Arj[e], i, pa, e);
    }
    const int nc = (int)cols.size();
    order.resize(nc);
    for (int q = 0; q < nc; q++) order[q] = q;
    std::sort(order.begin(), order.end(), [&](int x, int y) { return cols[x] < cols[y]; });
    newslot.resize(nc); start.resize(nc + 1);
    start[0] = 0;
    for (int q = 0; q < nc; q++) { newslot[order[q]] = q; start[q + 1] = start[q] + cnt[order[q]]; }
    const size_t tbase = trow.size();
    for (int q = 0; q < nc; q++) { Kj.push_back(cols[order[q]]); tptr.push_back((int)tbase + start[q]); }
    trow.resize(tbase + terms.size()); ta.resize(tbase + terms.size()); tb.resize(tbase + terms.size());
    for (const Term &t : terms) { const size_t o = tbase + (size_t)start[newslot[t.s]]++; trow[o] = t.row; ta[o] = t.a; tb[o] = t.b; }      // (stable: terms of an entry keep their order)
    Krp[j + 1] = (int)Kj.size();
  }
  tptr.push_back((int)trow.size());
  const int nzK = (int)Kj.size();
  std::vector<int> rb = build_row_blocks(Krp, n), runs;
  auto up_i = [&](const std::vector<int> &h) { int *p = dev_vec<int>(d_, h.size()); if (!h.empty()) be::h2d(d_, p, h.data(), sizeof(int) * h.size()); return p; };
  DevKf &f = d_.kf;
  DevCsr &K = f.K;
  K.nrows = n; K.ncols = n; K.nnz = nzK; K.nblk = (int)rb.size() - 1; K.split = n; K.nwin = 0;
  K.single = (int)rb.size() - 1 <= kGrid;
  for (size_t b = 0; b + 1 < rb.size(); b++) if (rb[b + 1] - rb[b] > kBlock) K.single = 0;
  K.rowptr = up_i(Krp); K.col = up_i(Kj); K.blkdesc = up_i(block_descs(rb, Krp, Kj, runs)); K.runinfo = up_i(runs);
  K.val = dev_vec<double>(d_, nzK);
  f.tptr = up_i(tptr); f.trow = up_i(trow); f.ta = up_i(ta); f.tb = up_i(tb); f.nterm = (int)trow.size();
  f.rec = dev_vec<double>(d_, 8 * (size_t)n);
  f.on = 1;
  if (pol_.setup_timing) std::fprintf(stderr, "[osqp_hip] K form: nnz(K) = %d (%.1f per row), %d terms, %d row blocks, %.1f ms\n", nzK, (double)nzK / n, f.nterm, K.nblk, 1e3 * (now_s() - t0));
}

// ------------------------------------------------------------------------------------------------ Woodbury plan
// backend.h DevWb: the rows of A with more than kLongRow entries, when there are between 1 and kWbMaxRows of them, are treated exactly
// in the preconditioner.  Symbolic data: the long rows as their own CSR (r x n), its transpose (n x r), where each entry sits in A.val.
void Engine::prepare_wb(const std::vector<int> &Arp, const std::vector<int> &Arj) {
  d_.wb = DevWb();
  if (!pol_.woodbury || !be::wb_supported() || settings.cg_precond != OSQP_DIAGONAL_PRECONDITIONER || m == 0) return;
  std::vector<int> rows;
  for (int i = 0; i < m; i++) if (Arp[i + 1] - Arp[i] > kLongRow) rows.push_back(i);
  const int r = (int)rows.size();
  if (r < 1) return;
  // many long rows: dense S on the device (backend.h kWbLargeMax) -- when the libraries load, the dense blocks fit comfortably (W, S, S^-1:
  // 8 (r ct + 2 r^2) bytes against a budget of 24 GiB of the 288) and the long rows carry most of A (else Jacobi is not the problem)
  bool large = false, dual = false;
  std::vector<int> colmap, cntL;
  int ct = 0, cd = 0;
  if (r > kWbMaxRows) {
    if (r > kWbLargeMax || !pol_.woodbury_large || !be::wb_large_supported()) return;
    size_t nz_long = 0;
    for (int i : rows) nz_long += (size_t)(Arp[i + 1] - Arp[i]);
    if (2 * nz_long < (size_t)Arp[m]) return;
    colmap.assign(n, -1);
    for (int i : rows) for (int k = Arp[i]; k < Arp[i + 1]; k++) colmap[Arj[k]] = 0;
    for (int j = 0; j < n; j++) if (colmap[j] == 0) colmap[j] = ct++;
    // column-space ("dual") form (backend.h DevWb::dual): the touched columns split into dense ones (two or more long-row entries) and singletons (one);
    // when the dense columns are fewer than the long rows the cd x cd system on them is the smaller one to form, factorise, invert and stream
    cntL.assign(n, 0);
    for (int i : rows) for (int k = Arp[i]; k < Arp[i + 1]; k++) cntL[Arj[k]]++;
    for (int j = 0; j < n; j++) cd += cntL[j] >= 2;
    dual = pol_.woodbury_dual != 0 && cd >= 1 && 4 * (long)cd <= 3 * (long)r;
    const double order = dual ? cd : r, width = dual ? cd : ct;
    if (8.0 * ((double)r * width + 2.0 * order * order) > 24.0 * 1024 * 1024 * 1024) return;
    large = true;
  }
  std::vector<unsigned char> islong(m, 0);
  std::vector<int> lrp(r + 1, 0), lcol, lsrc;
  for (int a = 0; a < r; a++) {
    const int i = rows[a]; islong[i] = 1;
    for (int k = Arp[i]; k < Arp[i + 1]; k++) { lcol.push_back(Arj[k]); lsrc.push_back(k); }
    lrp[a + 1] = (int)lcol.size();
  }
  std::vector<int> trp(n + 1, 0), tcol(lcol.size()), tsrc(lcol.size());
  for (int c : lcol) trp[c + 1]++;
  for (int j = 0; j < n; j++) trp[j + 1] += trp[j];
  { std::vector<int> cur(trp.begin(), trp.end() - 1);
    for (int a = 0; a < r; a++) for (int k = lrp[a]; k < lrp[a + 1]; k++) { const int pos = cur[lcol[k]]++; tcol[pos] = a; tsrc[pos] = lsrc[k]; } }
  auto up_i = [&](const std::vector<int> &h) { int *p = dev_vec<int>(d_, h.size()); if (!h.empty()) be::h2d(d_, p, h.data(), sizeof(int) * h.size()); return p; };
  auto up_csr = [&](DevCsr &M, int nr, int nc, const std::vector<int> &rp, const std::vector<int> &cj) {
    std::vector<int> rb;
    if (&M == &d_.wb.AL) { for (int a = 0; a <= nr; a++) rb.push_back(a); }      // every long row is a block of its own
    else rb = build_row_blocks(rp, nr);
    std::vector<int> runs;
    M.nrows = nr; M.ncols = nc; M.nnz = (int)cj.size(); M.nblk = (int)rb.size() - 1; M.split = nc; M.single = 0; M.nwin = 0;
    M.rowptr = up_i(rp); M.col = up_i(cj); M.blkdesc = up_i(block_descs(rb, rp, cj, runs)); M.runinfo = up_i(runs);
    M.val = dev_vec<double>(d_, cj.size());
  };
  DevWb &w = d_.wb;
  w.r = r;
  up_csr(w.AL, r, n, lrp, lcol); up_csr(w.ALT, n, r, trp, tcol);
  w.al_src = up_i(lsrc); w.alt_src = up_i(tsrc); w.rows = up_i(rows);
  w.islong = dev_vec<unsigned char>(d_, m); be::h2d(d_, w.islong, islong.data(), m);
  const size_t ord = dual ? (size_t)cd : (size_t)r;           // order of the dense system (backend.h DevWb: S / T)
  w.S = dev_vec<double>(d_, ord * ord); w.Sinv = dev_vec<double>(d_, ord * ord);
  w.g = dev_vec<double>(d_, r); w.h = dev_vec<double>(d_, r); w.Dinv0 = dev_vec<double>(d_, n);
  if (large && dual) {
    // dense columns get the positions 0 .. cd - 1 of W / T; every singleton column knows its long row and where its entry sits in A.val
    std::vector<int> kind(n, 0), dcol, srow(n, 0), ssrc(n, 0), sg_ptr(r + 1, 0), sg_col;
    for (int j = 0; j < n; j++) { colmap[j] = -1; if (cntL[j] >= 2) { kind[j] = 1; colmap[j] = (int)dcol.size(); dcol.push_back(j); } else if (cntL[j] == 1) kind[j] = 2; }
    for (int a = 0; a < r; a++) {
      for (int k = lrp[a]; k < lrp[a + 1]; k++) if (kind[lcol[k]] == 2) { srow[lcol[k]] = a; ssrc[lcol[k]] = lsrc[k]; sg_col.push_back(lcol[k]); }
      sg_ptr[a + 1] = (int)sg_col.size();
    }
    w.large = 1; w.dual = 1; w.cd = cd; w.ct = cd; w.colmap = up_i(colmap);
    w.kind = up_i(kind); w.dcol = up_i(dcol); w.srow = up_i(srow); w.ssrc = up_i(ssrc); w.sg_ptr = up_i(sg_ptr); w.sg_col = up_i(sg_col);
    w.sval = dev_vec<double>(d_, n); w.uz = dev_vec<double>(d_, n);
    w.wv = dev_vec<double>(d_, r); w.den = dev_vec<double>(d_, r); w.beta = dev_vec<double>(d_, r); w.wbeta = dev_vec<double>(d_, r); w.rt = dev_vec<double>(d_, r);
    w.W = dev_vec<double>(d_, (size_t)r * cd);                  // (zero-filled by the allocator: only the pattern's positions are ever written)
    w.pv = dev_vec<double>(d_, (size_t)n + m + n + 4 + r);
    // fused ADMM iteration of the direct mode (backend.h DevWb::fused): per-row tables now, the block views of A / B once those are uploaded (build_wbf_views)
    { std::vector<int> lidx(m, 0); for (int a = 0; a < r; a++) lidx[rows[a]] = a; w.lidx = up_i(lidx); }
    w.cc = dev_vec<double>(d_, m); w.sig = dev_vec<double>(d_, r);
    w.fused = (pol_.woodbury_fused != 0 && pol_.woodbury_direct != 0) ? 1 : 0;
    wb_kind_ = kind;
  } else if (large) {
    w.large = 1; w.ct = ct; w.colmap = up_i(colmap);
    w.W = dev_vec<double>(d_, (size_t)r * ct);                  // (zero-filled by the allocator: only the pattern's positions are ever written)
    w.pv = dev_vec<double>(d_, (size_t)n + m + n + 4 + r);
  } else w.WT = dev_vec<double>(d_, (size_t)n * r);
  w.info = dev_vec<int>(d_, 2); w.dbg = dev_vec<int>(d_, 1);
  if (large) { w.vendor = pol_.woodbury_vendor != 0; w.gjwork = dev_vec<double>(d_, wb_inverse_work((int)ord) + 1); }
  if (pol_.debug_fail_refactor > 0) { const int v = pol_.debug_fail_refactor; be::h2d(d_, w.dbg, &v, sizeof(int)); }
  w.on = 1;
  // K0 diagonal <=> P has diagonal entries only and every short row of A has exactly one entry: then M = K (backend.h DevWb::exact)
  bool diag = pol_.woodbury_direct != 0;
  // A long row that stores the same (row, column) twice (a valid CSC may: the engine keeps duplicates as separate CSR entries, and every SpMV
  // sums them) has ONE cell in the dense tiles / the dense transpose, which the fill kernels ASSIGN: the tile would hold one of the two
  // values.  As a preconditioner that is merely a slightly different M; as the direct mode (M taken for K) it would solve another system.
  { std::vector<int> mark(n, -1);
    for (int a = 0; a < r && diag; a++) for (int k = lrp[a]; k < lrp[a + 1]; k++) { if (mark[lcol[k]] == a) { diag = false; break; } mark[lcol[k]] = a; } }
  for (int j = 0; j < n && diag; j++) for (int k = P_.p[j]; k < P_.p[j + 1]; k++) if (P_.i[k] != j) { diag = false; break; }
  for (int i = 0; i < m && diag; i++) if (!islong[i] && Arp[i + 1] - Arp[i] > 1) diag = false;
  w.exact = diag ? 1 : 0;
  // (large mode: decided numerically after every factorisation -- two-entry rows whose contributions to K0's off-diagonal cancel, as in
  //  the lasso's  -t <= x <= t , are as good as one-entry rows)
  if (large) { w.probe = pol_.woodbury_direct != 0; w.exact = 0; w.log = pol_.woodbury_log; w.cache_on = pol_.woodbury_cache != 0; w.exact_tol = pol_.woodbury_direct_tol > 0 ? pol_.woodbury_direct_tol : 1e-6; }
  // The direct mode in two launches per ADMM iteration (backend.h DevWbx): additionally every short row has EXACTLY one entry (an empty
  // row would have no column to be updated with) and the problem is small enough for the per-workgroup partials (n <= kWbxMaxN)
  if (w.exact && !large && pol_.woodbury_fused && be::wbx_supported() && n <= kWbxMaxN) {
    bool ok = true;
    for (int i = 0; i < m && ok; i++) if (!islong[i] && Arp[i + 1] - Arp[i] != 1) ok = false;
    if (ok) {
      std::vector<int> sc_ptr(n + 1, 0), sc_row, sc_src;
      for (int j = 0; j < n; j++) {
        for (int k = A_.p[j]; k < A_.p[j + 1]; k++) if (!islong[A_.i[k]]) { sc_row.push_back(A_.i[k]); sc_src.push_back(AmapA_[k]); }
        sc_ptr[j + 1] = (int)sc_row.size();
      }
      DevWbx &x = w.x;
      x.G = (n + kWbxCols - 1) / kWbxCols; x.nsc = (int)sc_row.size();
      x.tile = dev_vec<double>(d_, (size_t)x.G * kWbMaxRows * kWbxCols);       // (zero-filled by the allocator: only the pattern's positions are ever written)
      x.tile2 = dev_vec<double>(d_, (size_t)x.G * kWbMaxRows * kWbxCols);
      x.partG = dev_vec<double>(d_, (size_t)x.G * kWbMaxRows); x.partZ = dev_vec<double>(d_, (size_t)x.G * kWbMaxRows);
      x.ls0 = dev_vec<double>(d_, 3 * (size_t)kWbMaxRows); x.ls1 = dev_vec<double>(d_, 3 * (size_t)kWbMaxRows);
      x.lz0 = dev_vec<double>(d_, 4 * (size_t)kWbMaxRows); x.lz1 = dev_vec<double>(d_, 4 * (size_t)kWbMaxRows); x.sinvp = dev_vec<double>(d_, (size_t)kWbMaxRows * kWbMaxRows);
      x.one = pol_.woodbury_fused == 1 ? 1 : 0;          // (OSQPHipPolicy::woodbury_fused = 2: the two-launch form of rounds 3-5, for A/B runs)
      x.sc_ptr = up_i(sc_ptr); x.sc_row = up_i(sc_row); x.sc_src = up_i(sc_src); x.sc_val = dev_vec<double>(d_, sc_row.size());
      x.bjj = dev_vec<double>(d_, n);
      be::wbx_init(d_);
      x.on = 1;
    }
  }
}

// Block views for the fused column-space iteration (backend.h DevWb::Bd / Bn / As): the same CSR arrays as d_.B / d_.A with a filtered block list --
// row blocks of B that hold a dense column, row blocks of B that hold a column that is not dense (a mixed block is in both: the epilogues filter by
// kind), row blocks of A that hold a short row.  A long row's descriptor keeps its offset into the run table.
void Engine::build_wbf_views(const std::vector<int> &rbA, const std::vector<int> &rbB, const std::vector<int> &Arp, const std::vector<int> &Arj, const std::vector<int> &Brp, const std::vector<int> &Bj) {
  DevWb &w = d_.wb;
  if (!(w.on && w.dual && w.fused)) return;
  std::vector<int> runs, dB = block_descs(rbB, Brp, Bj, runs), dA = block_descs(rbA, Arp, Arj, runs);
  std::vector<int> bd, bn, as;
  for (size_t b = 0; b + 1 < rbB.size(); b++) {
    bool dense = false, other = false;
    for (int j = rbB[b]; j < rbB[b + 1]; j++) { if (wb_kind_[j] == 1) dense = true; else other = true; }
    if (dense) bd.insert(bd.end(), dB.begin() + 4 * b, dB.begin() + 4 * b + 4);
    if (other) bn.insert(bn.end(), dB.begin() + 4 * b, dB.begin() + 4 * b + 4);
  }
  for (size_t b = 0; b + 1 < rbA.size(); b++) {
    bool shortrow = false;
    for (int i = rbA[b]; i < rbA[b + 1]; i++) if (Arp[i + 1] - Arp[i] <= kLongRow) shortrow = true;
    if (shortrow) as.insert(as.end(), dA.begin() + 4 * b, dA.begin() + 4 * b + 4);
  }
  auto up_i = [&](const std::vector<int> &h) { int *p = dev_vec<int>(d_, h.size()); if (!h.empty()) be::h2d(d_, p, h.data(), sizeof(int) * h.size()); return p; };
  auto view = [&](const DevCsr &M, const std::vector<int> &desc) { DevCsr V = M; V.blkdesc = up_i(desc); V.nblk = (int)desc.size() / 4; V.blkwin = nullptr; V.lcol = nullptr; V.nwin = 0; V.single = 0; return V; };
  w.Bd = view(d_.B, bd); w.Bn = view(d_.B, bn); w.As = view(d_.A, as);
  { int mx = 0;
    for (int j = 0; j < n; j++) if (wb_kind_[j] != 1) mx = std::max(mx, Brp[j + 1] - Brp[j]);
    for (int i = 0; i < m; i++) if (Arp[i + 1] - Arp[i] <= kLongRow) mx = std::max(mx, Arp[i + 1] - Arp[i]);
    w.thin = (pol_.woodbury_fused == 1 && mx <= 64) ? 1 : 0; }
  // the dense block held dense (backend.h DevWb::dense): OSQPHipPolicy::woodbury_fused = 2 keeps the CSR passes (A/B runs)
  long nzd = 0;
  for (int i = 0; i < m; i++) if (Arp[i + 1] - Arp[i] > kLongRow) for (int k = Arp[i]; k < Arp[i + 1]; k++) nzd += wb_kind_[Arj[k]] == 1;
  const size_t cells = (size_t)w.r * w.cd;
  if (pol_.woodbury_fused == 1 && w.cd % 2 == 0 && w.cd >= 512 && cells <= ((size_t)1 << 29) && 2 * (size_t)nzd >= cells) {
    std::vector<int> qp(w.cd + 1, 0), qi, qc;
    int c = 0;
    for (int j = 0; j < n; j++) if (wb_kind_[j] == 1) {
      for (int k = Brp[j]; k < Brp[j + 1]; k++) { const int col = Bj[k]; if (col >= n && Arp[col - n + 1] - Arp[col - n] > kLongRow) continue; qi.push_back(k); qc.push_back(col); }
      qp[++c] = (int)qi.size();
    }
    w.bq_ptr = up_i(qp); w.bq_idx = up_i(qi); w.bq_col = up_i(qc);
    w.grb = std::max(1, std::min(128, (w.r + 63) / 64));
    w.Ad = dev_vec<double>(d_, cells); w.ud = dev_vec<double>(d_, w.cd); w.ccd = dev_vec<double>(d_, w.r); w.gp = dev_vec<double>(d_, (size_t)w.grb * w.cd);
    w.dense = 1;
  }
}

// ------------------------------------------------------------------------------------------------ setup
int Engine::setup(const OSQPCscMatrix *P, const double *q, const OSQPCscMatrix *A, const double *l, const double *u,
                  int m_, int n_, const OSQPSettings *s) {
  double t0 = now_s();
  const bool ptime = pol_.setup_timing != 0;
  double tl = t0;
  auto lap = [&](const char *what) { if (ptime) { double t = now_s(); std::fprintf(stderr, "[osqp_hip setup] %-28s %8.2f ms\n", what, 1e3 * (t - tl)); tl = t; } };
  // ---- data validation (the C core's validate_data; error numbering bindings.cpp.in:364-375) ----
  if (!P || !A || !q || n_ <= 0 || m_ < 0) return OSQP_DATA_VALIDATION_ERROR;
  if (m_ > 0 && (!l || !u)) return OSQP_DATA_VALIDATION_ERROR;
  if (P->m != n_ || P->n != n_ || A->m != m_ || A->n != n_) return OSQP_DATA_VALIDATION_ERROR;
  auto csc_ok = [](const OSQPCscMatrix *M) {
    if (!M->p) return false;
    if (M->p[0] != 0) return false;
    for (int j = 0; j < M->n; j++) if (M->p[j + 1] < M->p[j]) return false;
    int nz = M->p[M->n];
    if (nz > 0 && (!M->i || !M->x)) return false;
    for (int k = 0; k < nz; k++) if (M->i[k] < 0 || M->i[k] >= M->m) return false;
    return true;
  };
  if (!csc_ok(P) || !csc_ok(A)) return OSQP_DATA_VALIDATION_ERROR;
  for (int j = 0; j < n_; j++)
    for (int k = P->p[j]; k < P->p[j + 1]; k++) if (P->i[k] > j) return OSQP_DATA_VALIDATION_ERROR;   // upper triangular only
  for (int i = 0; i < m_; i++) if (!(l[i] <= u[i])) return OSQP_DATA_VALIDATION_ERROR;
  int err = validate_settings(s, true);
  if (err) return err;

  free_all();
  n = n_; m = m_; settings = *s;
  rho_bar_ = clamp_rho(settings.rho); settings.rho = rho_bar_;                            // _osqp.py:503
  auto copy_csc = [](HostCsc &H, const OSQPCscMatrix *M) {
    H.nr = M->m; H.nc = M->n; int nz = M->p[M->n];
    H.p.assign(M->p, M->p + M->n + 1); H.i.assign(M->i, M->i + nz); H.x.assign(M->x, M->x + nz);
  };
  copy_csc(P_, P); copy_csc(A_, A);
  q0_.assign(q, q + n); l0_.assign(l, l + m); u0_.assign(u, u + m);
  lap("validate + copy");

  // ---- device ----
  err = be::init(d_, settings.device);
  if (err) return err;
  lap("device init");
  dev_ready_ = true;
  // ---- scaling: on the device once the matrices are assembled there; the test-only host simulator
  //      keeps the driver's host restatement of _osqp.py:389-497 ----
  const bool dev_asm = be::device_assembly();
  std::vector<double> Px, Ax, qs;
  d_.n = n; d_.m = m; d_.sigma = settings.sigma; d_.alpha = settings.alpha;

  const int nzA = A_.nnz(), nzP = P_.nnz();
  std::vector<int> Arp, Arj, Brp, Bj;
  int nzB = 0;
  // (a lambda: setup may build the structure twice -- as given, and for the reordered problem)
  auto build_structure = [&]() {
    // A as CSR (the incoming CSC is CSR(A')) + map CSC index -> CSR position
    Arp.assign(m + 1, 0); Arj.assign(nzA, 0);
    AmapA_.resize(nzA);
    for (int k = 0; k < nzA; k++) Arp[A_.i[k] + 1]++;
    for (int i = 0; i < m; i++) Arp[i + 1] += Arp[i];
    {
      std::vector<int> cur(Arp.begin(), Arp.end() - 1);
      for (int j = 0; j < n; j++)
        for (int k = A_.p[j]; k < A_.p[j + 1]; k++) { int pos = cur[A_.i[k]]++; Arj[pos] = j; AmapA_[k] = pos; }
    }
    // B = [P + sigma I | A'] as CSR with n rows; row j = (lower part of row j of P) (diag) (upper part) (column j of A)
    Brp.assign(n + 1, 0);
    std::vector<char> hasdiag(n, 0);
    for (int j = 0; j < n; j++) {
      Brp[j + 1] += 1 + (A_.p[j + 1] - A_.p[j]);
      for (int k = P_.p[j]; k < P_.p[j + 1]; k++) {
        int i = P_.i[k];
        if (i == j) hasdiag[j] = 1; else { Brp[j + 1]++; Brp[i + 1]++; }
      }
    }
    for (int j = 0; j < n; j++) Brp[j + 1] += Brp[j];
    nzB = Brp[n];
    Bj.assign(nzB, 0); std::vector<int> &bdiag = bdiag_; bdiag.assign(n, 0);
    Pmap1_.assign(nzP, -1); Pmap2_.assign(nzP, -1); AmapB_.resize(nzA);
    {
      std::vector<int> cur(Brp.begin(), Brp.end() - 1);
      for (int j = 0; j < n; j++) {
        for (int k = P_.p[j]; k < P_.p[j + 1]; k++) {
          int i = P_.i[k];
          if (i == j) continue;
          int p1 = cur[j]++; Bj[p1] = i; Pmap1_[k] = p1;      // (j, i): lower part of row j
        }
        bdiag[j] = cur[j]++; Bj[bdiag[j]] = j;
        for (int k = P_.p[j]; k < P_.p[j + 1]; k++)           // every stored (j, j) entry -- valid CSC may repeat it -- adds into the one slot
          if (P_.i[k] == j) Pmap1_[k] = bdiag[j];
        for (int k = P_.p[j]; k < P_.p[j + 1]; k++) {
          int i = P_.i[k];
          if (i == j) continue;
          int p2 = cur[i]++; Bj[p2] = j; Pmap2_[k] = p2;      // (i, j): upper part of row i (its diagonal is already placed)
        }
      }
      for (int j = 0; j < n; j++)
        for (int k = A_.p[j]; k < A_.p[j + 1]; k++) { int pos = cur[j]++; Bj[pos] = n + A_.i[k]; AmapB_[k] = pos; }
    }
  };
  build_structure();
  lap("CSR(A), B structure, maps");
  std::vector<int> rbA = build_row_blocks(Arp, m), rbB = build_row_blocks(Brp, n);
  lap("row blocks");
  // One launch per PCG iteration (F1 form): wants row blocks of A of at most kF1Chunk entries -- on large problems (n = 1M: the default
  // blocks hold ~2000 entries) A is re-blocked for it, a workgroup then loops over several blocks per launch; when the plan does not
  // apply the default blocks stay
  bool has_long = false;
  for (int i = 0; i < m && !has_long; i++) has_long = Arp[i + 1] - Arp[i] > kLongRow;
  const bool want_f1 = pol_.pcg_fused && use_slots_ && pol_.window != 0 && pol_.f1 && !has_long;
  F1Plan plan;
  auto try_plan = [&]() {
    if (!want_f1) return false;
    if (plan_f1(rbA, Arp, Arj, Brp, Bj, plan)) return true;
    if ((long)nzA > (long)kGrid * kF1Chunk) {
      std::vector<int> rbF = build_row_blocks(Arp, m, kF1Chunk);
      if (plan_f1(rbF, Arp, Arj, Brp, Bj, plan)) { rbA.swap(rbF); return true; }
    } else {
      // mid-size problems: the default blocking spreads A over all kGrid workgroups (n = 50k: 1024 blocks of ~490 entries) -- blocks half as
      // tall as the band is wide need more than kF1MaxD replicas.  FULL blocks (fewer than kGrid of them: some workgroups idle in the F
      // launches) keep the one-launch form applicable down to a quarter of the grid
      std::vector<int> rbF = build_row_blocks_target(Arp, m, kF1Chunk - 24);
      if ((int)rbF.size() - 1 >= kGrid / 4 && plan_f1(rbF, Arp, Arj, Brp, Bj, plan)) { rbA.swap(rbF); return true; }
    }
    return false;
  };
  bool f1ok = try_plan();
  // Reordering (OSQPHipPolicy::reorder; Engine::compute_reorder): 1 = when the one-launch form does not apply to the problem as given,
  // look for a permutation under which it does and keep it only then -- never for a QP small enough for the batch kernel, which takes the
  // caller's numbering only (a handle that serves batches keeps doing so); 2 = always work on the permuted problem (tests of the plumbing)
  clear_reorder(); reorder_ms_ = 0;
  const int reorder = no_reorder_ ? 0 : pol_.reorder;
  if (m > 0 && (reorder == 2 || (reorder == 1 && want_f1 && !f1ok && be::device_assembly() && (int)rbA.size() - 1 >= kGrid / 4 && !be::batch_lds_bytes(n, m)))) {
    const double tr = now_s();
    if (compute_reorder(Arp, Arj, Brp, Bj) || reorder == 2) {
      if (pc_.empty()) {                             // (forced mode on a graph the search gave up on: the identity permutation exercises the plumbing just as well)
        pc_.resize(n); ipc_.resize(n); pr_.resize(m); ipr_.resize(m);
        for (int j = 0; j < n; j++) pc_[j] = ipc_[j] = j;
        for (int i = 0; i < m; i++) pr_[i] = ipr_[i] = i;
      }
      HostCsc P0 = P_, A0 = A_; std::vector<double> q00 = q0_, l00 = l0_, u00 = u0_;
      apply_reorder();
      build_structure();
      rbA = build_row_blocks(Arp, m); rbB = build_row_blocks(Brp, n);
      f1ok = try_plan();
      if (!f1ok && reorder != 2) {                   // no gain: the problem stays as the caller numbered it
        P_ = std::move(P0); A_ = std::move(A0); q0_ = std::move(q00); l0_ = std::move(l00); u0_ = std::move(u00);
        clear_reorder();
        build_structure();
        rbA = build_row_blocks(Arp, m); rbB = build_row_blocks(Brp, n);
        f1ok = try_plan();
      }
    }
    reorder_ms_ = 1e3 * (now_s() - tr);
    lap("reordering");
  }
  if (!dev_asm) { Px = P_.x; Ax = A_.x; qs = q0_; compute_scaling(Px, Ax, qs); lap("Ruiz scaling (host)"); }
  Arp_ = Arp; Arj_ = Arj; Brp_ = Brp; Bj_ = Bj;
  d_.fused = pol_.pcg_fused ? 1 : 0;                 // 0 selects the 3-kernel sequence
  prepare_wb(Arp, Arj);
  if (d_.wb.on) d_.fused = 0;                        // (the Woodbury-corrected preconditioner lives in the three-kernel PCG form)
  d_.f1 = DevF1();
  if (d_.fused && f1ok) upload_f1(plan);
  // (neither the one-launch form on A alone nor a Woodbury mode: the explicit reduced matrix, where its fill is moderate -- backend.h DevKf)
  if (d_.fused && !d_.f1.on && use_slots_) prepare_kf(Arp, Arj, Brp, Bj);
  if (reordered_) {
    d_pc_ = dev_vec<int>(d_, n); d_pr_ = dev_vec<int>(d_, m);
    be::h2d(d_, d_pc_, pc_.data(), sizeof(int) * n); be::h2d(d_, d_pr_, pr_.data(), sizeof(int) * m);
  }
  lap("F1 / Woodbury plans");
  // block descriptors; long rows also get their run table (see DevCsr::runinfo).  Slices are the fixed kChunk steps the kernels
  // take from the row's first entry (cutting them at run starts instead adds short slices that cost more than the saved
  // index bytes: lasso PCG pair 208 us vs 220 us)
  auto descs = [](const std::vector<int> &rb, const std::vector<int> &rp, const std::vector<int> &cj, std::vector<int> &runs) {
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
  };

  // column windows of the (short-row) blocks, see DevCsr::blkwin.  OSQPHipPolicy::window = 0 turns the windowed path off (A/B runs).
  const bool win_on = pol_.window != 0;
  auto windows = [win_on](const std::vector<int> &rb, const std::vector<int> &rp, const std::vector<int> &cj, int split,
                    std::vector<int> &win, std::vector<unsigned short> &lcol) {
    const size_t nb = rb.size() - 1;
    win.assign(4 * nb, 0); lcol.assign(std::max<size_t>(cj.size(), 1), 0);
    int nwin = 0;
    for (size_t b = 0; b < nb; b++) {
      const int r0 = rb[b], r1 = rb[b + 1], k0 = rp[r0], k1 = rp[r1];
      win[4 * b + 1] = -1;
      if (!win_on || (r1 - r0 == 1 && k1 - k0 > kLongRow) || k1 == k0) continue;
      int lo0 = INT32_MAX, hi0 = -1, lo1 = INT32_MAX, hi1 = -1;
      for (int k = k0; k < k1; k++) {
        const int c = cj[k];
        if (c < split) { lo0 = std::min(lo0, c); hi0 = std::max(hi0, c); } else { lo1 = std::min(lo1, c - split); hi1 = std::max(hi1, c - split); }
      }
      const long len0 = hi0 >= 0 ? (long)hi0 - lo0 + 1 : 0, len1 = hi1 >= 0 ? (long)hi1 - lo1 + 1 : 0;
      if (len0 + len1 > kWinCap) continue;
      if (len0 == 0) lo0 = 0;
      if (len1 == 0) lo1 = 0;
      win[4 * b] = lo0; win[4 * b + 1] = (int)len0; win[4 * b + 2] = lo1; win[4 * b + 3] = (int)len1;
      for (int k = k0; k < k1; k++) {
        const int c = cj[k];
        lcol[k] = (unsigned short)(c < split ? c - lo0 : len0 + (c - split - lo1));
      }
      nwin++;
    }
    return nwin;
  };
  auto up_i = [&](const std::vector<int> &h) { int *p = dev_vec<int>(d_, h.size()); be::h2d(d_, p, h.data(), sizeof(int) * h.size()); return p; };
  auto up_win = [&](DevCsr &M, const std::vector<int> &rb, const std::vector<int> &rp, const std::vector<int> &cj, int split) {
    std::vector<int> win; std::vector<unsigned short> lcol;
    M.split = split;
    M.single = (int)rb.size() - 1 <= kGrid;
    for (size_t b = 0; b + 1 < rb.size(); b++) if (rb[b + 1] - rb[b] > kBlock) M.single = 0;
    M.nwin = windows(rb, rp, cj, split, win, lcol);
    M.blkwin = up_i(win);
    M.lcol = dev_vec<unsigned short>(d_, lcol.size());
    be::h2d(d_, M.lcol, lcol.data(), sizeof(unsigned short) * lcol.size());
  };
  d_.A.nrows = m; d_.A.ncols = n; d_.A.nnz = nzA; d_.A.nblk = (int)rbA.size() - 1;
  d_.A.rowptr = up_i(Arp); d_.A.col = up_i(Arj); { std::vector<int> runs; d_.A.blkdesc = up_i(descs(rbA, Arp, Arj, runs)); d_.A.runinfo = up_i(runs); } d_.A.val = dev_vec<double>(d_, nzA);
  d_.B.nrows = n; d_.B.ncols = n + m; d_.B.nnz = nzB; d_.B.nblk = (int)rbB.size() - 1;
  d_.B.rowptr = up_i(Brp); d_.B.col = up_i(Bj); { std::vector<int> runs; d_.B.blkdesc = up_i(descs(rbB, Brp, Bj, runs)); d_.B.runinfo = up_i(runs); } d_.B.val = dev_vec<double>(d_, nzB);
  d_.Bdiag = up_i(bdiag_);
  up_win(d_.A, rbA, Arp, Arj, n); up_win(d_.B, rbB, Brp, Bj, n);
  build_wbf_views(rbA, rbB, Arp, Arj, Brp, Bj);
  lap("upload structure");
  auto dv = [&](size_t cnt) { return dev_vec<double>(d_, cnt); };
  d_.q = dv(n); d_.l = dv(m); d_.u = dv(m); d_.D = dv(n); d_.Dinv = dv(n); d_.E = dv(m); d_.Einv = dv(m);
  d_.rho = dv(m); d_.rho_inv = dv(m); d_.ctype = dev_vec<int>(d_, m);
  d_.x = dv(n); d_.z = dv(m); d_.y = dv(m); d_.dx = dv(n); d_.dy = dv(m); d_.zt = dv(m); d_.t0 = dv(m); d_.v = dv(m);
  d_.xg = dv(n); d_.xsp = dv(n); d_.ztg = dv(m);
  d_.theta = pol_.extrap;                            // PCG start extrapolation (backend.h Dev::xg)
  { // write-through result stores of the two-kernel form (Dev::wt): while A, B and the vectors of an iteration stay in the 256 MiB Infinity Cache
    const double mb = (12.0 * ((double)d_.A.nnz + (double)d_.B.nnz) + 8.0 * (14.0 * n + 12.0 * m)) / (1024.0 * 1024.0);
    const char *ev = std::getenv("OSQP_HIP_WT");
    d_.wt = ev ? (ev[0] != '0') : (mb <= 160.0);
  }
  d_.uu = dv(n); d_.w = dv(n); d_.t = dv(m); d_.uu2 = dv(n); d_.ms = dv(2 * (size_t)n);
  if (d_.f1.on) { const size_t ns = d_.f1.ns; double *va = d_.f1.va; d_.Minv = va; d_.xs = va + ns; d_.p = va + 2 * ns; d_.r = va + 3 * ns; d_.s = va + 5 * ns; }   // backend.h DevF1::va
  else { d_.r = dv(n); d_.p = dv(n); d_.s = dv(n); d_.Minv = dv(n); d_.xs = dv(n); }
  d_.part = dv((size_t)kPartSlots * kGrid); d_.res = dv(R_COUNT); d_.scal = dv(S_HIST + 3 * (kMaxCg + 1)); d_.flags = dev_vec<int>(d_, F_COUNT); d_.slot = dev_vec<int>(d_, be::kSlotInts);
  d_.ctl = be::device_assembly() ? static_cast<Ctl *>(be::alloc(d_, sizeof(Ctl))) : nullptr;      // (the host simulator processes every boundary on the host)
  if (dev_asm) {
    // the caller's values go up once, in their own (CSC) order; every later (re)assembly and the equilibration run on the device
    std::vector<int> Pj(nzP), Aj(nzA);
    for (int j = 0; j < n; j++) { for (int k = P_.p[j]; k < P_.p[j + 1]; k++) Pj[k] = j; for (int k = A_.p[j]; k < A_.p[j + 1]; k++) Aj[k] = j; }
    d_.nzP = nzP; d_.nzA = nzA;
    d_.Praw = dv(nzP); d_.Araw = dv(nzA); d_.cs = dv(2);
    d_.Pi = up_i(P_.i); d_.Pj = up_i(Pj); d_.Pm1 = up_i(Pmap1_); d_.Pm2 = up_i(Pmap2_);
    d_.Ai = up_i(A_.i); d_.Aj = up_i(Aj); d_.AmA = up_i(AmapA_); d_.AmB = up_i(AmapB_);
    be::h2d(d_, d_.Praw, P_.x.data(), sizeof(double) * nzP); be::h2d(d_, d_.Araw, A_.x.data(), sizeof(double) * nzA);
    be::h2d(d_, d_.q, q0_.data(), sizeof(double) * n);
    be::assemble(d_, 0, 1.0, 0);                                     // unscaled, sigma added after the equilibration
    c_ = be::ruiz(d_, settings.scaling);                             // _osqp.py:389-497
    cinv_ = 1.0 / c_;
    be::f1_refresh(d_); be::wb_refresh(d_); be::wbx_refresh(d_);
    D_.resize(n); E_.resize(m); Dinv_.resize(n); Einv_.resize(m);
    be::d2h(d_, D_.data(), d_.D, sizeof(double) * n); be::d2h(d_, Dinv_.data(), d_.Dinv, sizeof(double) * n);
    if (m > 0) { be::d2h(d_, E_.data(), d_.E, sizeof(double) * m); be::d2h(d_, Einv_.data(), d_.Einv, sizeof(double) * m); }
    lap("assembly + Ruiz scaling (device)");
  } else {
    Aval_.assign(nzA, 0.0); Bval_.assign(nzB, 0.0);
    fill_matrix_values(Px, Ax);
    be::h2d(d_, d_.D, D_.data(), sizeof(double) * n); be::h2d(d_, d_.Dinv, Dinv_.data(), sizeof(double) * n);
    be::h2d(d_, d_.E, E_.data(), sizeof(double) * m); be::h2d(d_, d_.Einv, Einv_.data(), sizeof(double) * m);
    lap("matrix values (host-scaled)");
  }
  d_.qraw = dv(n); d_.lraw = dv(m); d_.uraw = dv(m); d_.cnt = dev_vec<int>(d_, 2);
  raw_stale_ = scaled_stale_ = false;
  if (be::device_vec_updates()) {
    be::copy_in(d_, d_.qraw, q0_.data(), sizeof(double) * n, 0);
    be::copy_in(d_, d_.lraw, l0_.data(), sizeof(double) * m, 0); be::copy_in(d_, d_.uraw, u0_.data(), sizeof(double) * m, 0);
    device_scale_vectors(true, true);
  } else {
    upload_q();
    upload_bounds_and_types();
  }
  be::set_rho(d_, rho_bar_);                                       // _osqp.py:499-524
  try { be::precond(d_, settings.cg_precond == OSQP_DIAGONAL_PRECONDITIONER); }
  catch (const DeviceError &err) {
    // the first device-side factorisation of the large-rank correction failed (a dense-library call, not this engine's kernels): the
    // handle falls back to plain Jacobi -- said loudly, and visible in OSQPHipStats::woodbury_rows = 0
    if (!(d_.wb.on && d_.wb.large)) throw;
    std::fprintf(stderr, "osqp_hip: large-rank Woodbury correction switched off for this handle (%s)\n", err.what());
    d_.wb.on = 0;
    be::precond(d_, settings.cg_precond == OSQP_DIAGONAL_PRECONDITIONER);
  }
  be::init_iterates(d_, 1);

  sol_x_.assign(n, kNaN); sol_y_.assign(m, kNaN); sol_pc_.assign(m, kNaN); sol_dc_.assign(n, kNaN);
  solution.x = sol_x_.data(); solution.y = sol_y_.data(); solution.prim_inf_cert = sol_pc_.data(); solution.dual_inf_cert = sol_dc_.data();
  std::memset(&info, 0, sizeof(info));
  set_status(OSQP_UNSOLVED);
  cg_budget_ = 0; have_tol_ = false; first_run_ = true; slot_pred_[0] = slot_pred_[1] = 6.0; slot_pred_[2] = 14.0;
  stats_ = OSQPHipStats(); stats_.nnzA = nzA; stats_.nnzB = nzB;
  be::sync(d_);
  lap("vectors, rho, preconditioner");
  info.setup_time = now_s() - t0;
  if (settings.verbose) print_setup_header();
  return OSQP_NO_ERROR;
}

// _osqp.py:564-607, with this engine's banner and linear-system line
void Engine::print_setup_header() const {
  say("-----------------------------------------------------------------\n");
  say("           OSQP v%s  -  Operator Splitting QP Solver\n", "1.0.0-hip");
  say("        ADMM engine for AMD MI355X (%s), C ABI of osqp v1\n", be::name());
  say("-----------------------------------------------------------------\n");
  say("problem:  variables n = %d, constraints m = %d\n", n, m);
  say("          nnz(P) + nnz(A) = %d\n", P_.nnz() + A_.nnz());
  say("settings: linear system solver = indirect (reduced-KKT PCG, %s),\n", settings.cg_precond == OSQP_DIAGONAL_PRECONDITIONER ? "diagonal preconditioner" : "no preconditioner");
  say("          eps_abs = %.2e, eps_rel = %.2e,\n", settings.eps_abs, settings.eps_rel);
  say("          eps_prim_inf = %.2e, eps_dual_inf = %.2e,\n", settings.eps_prim_inf, settings.eps_dual_inf);
  say("          rho = %.2e %s\n", settings.rho, settings.adaptive_rho ? "(adaptive)" : "");
  say("          sigma = %.2e, alpha = %.2f, max_iter = %d\n", settings.sigma, settings.alpha, settings.max_iter);
  say("          check_termination: %s (interval %d), cg_max_iter = %d\n", settings.check_termination ? "on" : "off", settings.check_termination, settings.cg_max_iter);
  say("          scaling: %s, scaled_termination: %s\n", settings.scaling ? "on" : "off", settings.scaled_termination ? "on" : "off");
  say("          warm_start: %s, polish: %s\n\n", settings.warm_starting ? "on" : "off", settings.polishing ? "on" : "off");
}

}  // namespace osqp_hip
