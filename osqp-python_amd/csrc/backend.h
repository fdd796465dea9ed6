// backend.h -- device-op interface between the host ADMM driver (engine.cpp) and the compute backend.
//
// Two implementations exist:
//   backend_hip.hip   hand-written gfx950 kernels; the ONLY backend linked into libosqp_hip.so (the product)
//   backend_host.cpp  plain loops with identical semantics; built ONLY by tests/ into tests/_build/ so that the
//                     driver logic (termination, adaptive rho, PCG budget, updates) can be exercised on machines
//                     without a GPU.  It is never shipped, never loaded by the package, and is not a fallback.
//
// Every ADMM quantity lives in SCALED space (Appendix A of SURVEY.md); formulas cite
// /root/reference/src/osqppurepy/_osqp.py ("_osqp.py:LINE").
//
// Matrices on the device (fp64 values, int32 indices):
//   A : m x n      CSR of the scaled constraint matrix
//   B : n x (n+m)  CSR of [ P + sigma I | A' ]  (P full symmetric; column n+i is constraint i)
// so that the reduced KKT operator of the PCG  K = P + sigma I + A' diag(rho) A  is applied as
//   t = rho .* (A p)   (kernel K1, m rows)      Kp = B [p; t]   (kernel K2, n rows).
#pragma once
#include <cstddef>
#include <cstdint>
#include <stdexcept>

namespace osqp_hip {

// A failed HIP runtime call.  Thrown by the backend, caught at the C-API boundary (api.cpp) and reported as an
// osqp_error_type value -- the library never calls abort() and never falls back to a CPU path.
struct DeviceError : std::runtime_error { using std::runtime_error::runtime_error; };

#ifndef OSQP_HIP_KBLOCK
#define OSQP_HIP_KBLOCK 256
#define OSQP_HIP_KGRID 1024
#endif
constexpr int kBlock = OSQP_HIP_KBLOCK;   // threads per workgroup (wave64s)
constexpr int kGrid = OSQP_HIP_KGRID;     // workgroups per launch = number of partial-reduction slots
constexpr int kChunk = 8 * kBlock;        // nnz staged through LDS per row-block (8 per thread, all loads in flight at once)
constexpr int kLongRow = 128;      // rows with more nnz get a workgroup of their own (block-wide reduction)
constexpr int kMaxRowsPerBlock = 1024;
constexpr int kPartSlots = 40;     // rows of Dev::part (each kGrid doubles)
constexpr int kMaxCg = 1024;       // hard cap on the PCG budget (size of the alpha/gamma history)
constexpr int kWinCap = 1472;      // input-vector window of a row block staged in LDS (elements; 23 KB as 16-byte pairs: 4 workgroups per CU)
// One-launch PCG iteration ("F1" form, DevF1 below)
constexpr int kF1Win = 512;        // widest window of a row block of A the F1 form takes (two elements per lane)
constexpr int kF1MaxD = 4;         // most replicas of the partial A' t vector
constexpr int kF1PChunk = 256;     // most (P + sigma I) entries of a block's own rows (one per lane)
constexpr int kF1MaxOwn = 512;     // most own columns of a block
constexpr int kF1MaxRows = 512;    // most rows of a block
constexpr int kF1Chunk = 1024;     // most entries of a row block of A
constexpr int kF1StreamBytes = kF1Chunk * 12;   // a block's slice of DevF1::stream: 8-byte values + 4-byte packed index words

struct DevCsr {
  int nrows = 0, ncols = 0, nnz = 0, nblk = 0;
  int *rowptr = nullptr, *col = nullptr;
  int *blkdesc = nullptr;        // nblk x {first row, end row, first nnz, end nnz}: one 16-byte load per row block.  A LONG row
                                 // (one row, more than kLongRow entries: reduced by the whole workgroup) carries -(1 + base) as its
                                 // end row: base = its first entry in runinfo
  int *runinfo = nullptr;        // per kChunk-entry slice of a long row: first column if the slice's columns are consecutive
                                 // (dense data blocks: the kernels then skip the index loads, 4 of 12 bytes per entry), else -1
  double *val = nullptr;
  // Windowed row blocks (banded matrices): the columns a block touches lie in at most two short ranges -- [w.x, w.x + w.y) of the
  // columns below `split` and [split + w.z, split + w.z + w.w) of those from `split` on (B = [P | A']: split = n; A: split = ncols) --
  // with w.y + w.w <= kWinCap.  The kernels then stage that window of the input vector(s) in LDS with coalesced loads and gather
  // from LDS through 16-bit LOCAL indices (10 instead of 12 bytes per entry, and no dependent global gather).
  int *blkwin = nullptr;         // nblk x {w.x, w.y, w.z, w.w}; w.y < 0: not windowed (global gather through col)
  unsigned short *lcol = nullptr;// [nnz] position inside the block's window (first range, then second); unused for other blocks
  int split = 0;
  int nwin = 0;                  // number of windowed blocks (statistics)
  int single = 0;                // 1: at most kGrid row blocks, each with at most kBlock rows (a workgroup's first pass covers all its rows)
};

// One launch per PCG iteration ("F1" form).  K = P + sigma I + A' diag(rho) A is applied from A ALONE: the
// workgroup that owns row block g of A forms  t = rho .* (A u)  for its rows and, in the same launch, the products  A_g' t_g  of those
// rows summed per column of the block's window (a second, column-ordered pass over the block's entries held in registers: the
// entries' 16-bit local row and their position in column order are stored next to the local column).  That partial slice goes to
// one of D "replica" n-vectors (block g -> replica g mod D; the windows of blocks g and g + D never overlap, and a block also zero-
// fills the gap up to the next window of its replica, so every replica is rewritten completely by every launch); whoever needs
// (A' t)_j next -- every workgroup whose window holds column j -- sums the D replicas in index order (deterministic, no atomics).
// The (P + sigma I) part is applied by the block to its OWN columns [cs0, cs1) from a compact CSR of P + sigma I.  One launch then
// performs a whole Chronopoulos-Gear iteration: scalars from the previous launch's partials, vector update on the window (each
// workgroup recomputes u_{k+1} on the columns it gathers -- nothing is exchanged inside a launch), SpMV, partials for the next.
struct DevF1 {
  int on = 0;                    // plan valid (Engine::prepare_f1); 0: the two-kernel form
  int D = 0;                     // replicas
  int *blk = nullptr;            // nblk(A) x 16 words, one scalar load per block:
                                 //   {first row, end row, first entry, end entry}  (the block's descriptor in A)
                                 //   {cov0, cov1, cs0, cs1}: replica coverage [cov0, cov1) (zero outside the window), own columns [cs0, cs1)
                                 //   {offset of the block's column pointers in cptr, first and end entry of its own rows in the P arrays, far columns (mix)}
                                 //   {g0, gl, a0, wl}: gather window [g0, g0 + gl) (columns of the block's rows of A -- plus those of its own
                                 //   rows of P if that widens it by at most a quarter), scatter window [a0, a0 + wl) (columns of its rows of A)
  // The matrix stream of a block, in the layout the kernel's LDS buffer has (pcg_hip.hip F1Stream): block b owns the kF1StreamBytes bytes at
  // stream + b * kF1StreamBytes = { double val[kF1Chunk]; unsigned ent[kF1Chunk]; } -- the block's values of A (copied from A.val by be::f1_refresh)
  // and one packed word per entry: column relative to g0 (9 bits) | local row << 9 (9 bits) | position of the entry in the block's
  // column-major order << 18 (11 bits).  Fixed stride: the address depends on the block index alone, so a workgroup requests its first block's
  // stream at the very head of a launch -- before any record has arrived -- and the NEXT block's stream while the current one is in its LDS
  // phases, both as LDS-direct loads (global_load_lds_dwordx4: no VGPR destination).
  unsigned char *stream = nullptr;
  unsigned short *cptr = nullptr;// per block: window length + 1 column pointers (block-local, column-major entry positions)
  int *prp = nullptr, *pcol = nullptr, *psrc = nullptr;   // compact CSR of P + sigma I (n rows): row pointers, columns, position of each entry in B.val
  double *pval = nullptr;        // values, refreshed from B.val by be::f1_refresh (after assembly / equilibration / matrix updates)
  int pnnz = 0;
  // the n-vectors of the iteration in ONE arena, stride ns doubles: Minv, x~, p, r (parity 0 = Dev::r, 1), s (0 = Dev::s, 1),
  // rep (parity 0: D vectors, parity 1: D vectors; their sum is K u_k: a block adds (P + sigma I) u_k of its own columns to its slice), and a third
  // set of D vectors: KA leaves the slices of rhs = sigma x - q + A' v in the parity-1 set and those of K x_g in the third (pcg_hip.hip f1_ka_body).
  // Dev::Minv / xs / p / r / s point into it.  r_k, s_{k-1} and rep_k (what launch F_k writes) live in parity k & 1.
  double *va = nullptr; size_t ns = 0;
  // Per-block mixing (mix = 1): a row block of A whose columns do not fit one window keeps a window of at most kF1Win - kF1MaxFar columns and
  // treats the remaining columns -- at most kF1MaxFar distinct ones -- as FAR columns.  They occupy the LAST kF1MaxFar slots of the block's gather
  // list (the same 4 + D vector loads at column fcol[.] instead of g0 + e; which lanes serve them does not depend on the block's record) and
  // nfc more segments of its column-ordered pass.  A far column's sum goes to the block's SPILL SLOT for it instead of a replica: spill is
  // ordered by (column, block), so whoever reconstructs column c adds its slots in index order to the replicas' sum -- written by launch F_k,
  // read by F_{k+1}, three sets like the replicas (parity 0, parity 1 / r_0's slices, rhs's slices).  No atomics, fixed order: results do not
  // depend on scheduling, and every workgroup that recomputes a column obtains the same bits.  blk word 11 = the block's number of far columns.
  int mix = 0;
  int wt = 0;                    // the launches' results are written through the XCD's L2 (pcg_hip.hip gst_): set while the working set fits the Infinity Cache
  int *fcol = nullptr, *fq = nullptr;   // fixed stride kF1MaxFar per block: far columns (ascending) as pairs {column, spk[column]}, and the spill slot of each
  int *sp_ptr = nullptr;         // [n + 1] spill slots by column (host-side order; kept for inspection)
  int *spk = nullptr;            // [n] one packed word per column: first slot << 6 | count  (the plan refuses columns with more than 63 slots)
  double *spill = nullptr; size_t nsp = 0;      // 3 x (nsp + 2): the sets are padded (a reader takes its first two slots unconditionally)
};
constexpr int kF1MaxFar = 128;     // most far columns of a row block

// One launch per PCG iteration for ANY sparsity pattern ("K form", pcg_hip.hip k_slotk).  The reduced operator of the PCG,
//     K = P + sigma I + A' diag(rho) A            (/root/reference/src/osqppurepy/_osqp.py:291-301 is the KKT matrix it is the Schur complement of)
// is held EXPLICITLY as a CSR matrix (n x n, full symmetric pattern = pattern(P) + union over the rows i of A of cols(i) x cols(i)), so that one
// PCG iteration is ONE sparse product instead of two dependent ones (t = rho .* (A u), then B [u; t]: two launches, each paying the launch boundary).
// Symbolic part, once at setup (Engine::prepare_kf): every K entry owns a list of TERMS -- a copy of an entry of B.val (the P + sigma I part,
// row < 0) or a product  rho_i A_ia A_ib  (row i, positions a, b in A.val); numeric part, k_kf_values: one streaming pass over the term lists
// wherever rho or the matrix values change (be::precond), i.e. a rho update costs one more small kernel, not a re-assembly.  Chosen at setup when
// neither the one-launch form on A alone (DevF1: banded A, 0.72 x the bytes) nor a Woodbury mode applies and the fill stays moderate
// (sum_i nnz(row i)^2 <= kKfMaxFill nnz(A): configs[1] with unstructured columns: 41 entries per row, 49 MB; lasso / portfolio rows are dense: never).
// The Chronopoulos-Gear recurrences fit the launch boundary as in the F1 form (pcg_hip.hip): launch F_k folds the previous launch's partials (gamma,
// delta, ||r||), rebuilds  u_k[c] = Minv (r_{k-1} - alpha (w_{k-1} + beta s_{k-2}))[c]  at EVERY column c it gathers from the column's 32-byte
// RECORD {Minv, r_{k-1}, w_{k-1}, s_{k-2}} (one aligned 32-byte gather per entry instead of four scattered ones; same instruction sequence as the
// owner's update: all copies are bit-identical), updates its own rows' x~, p, r, s, applies K and leaves the next record + partials.  Records are
// double-buffered by the parity of k.
constexpr int kKfMaxFill = 8;
struct DevKf {
  int on = 0;
  DevCsr K;
  int *tptr = nullptr;           // [nnz(K) + 1] term ranges
  int *trow = nullptr, *ta = nullptr, *tb = nullptr;   // per term: constraint row (< 0: copy of B.val[ta]), positions of A_ia, A_ib in A.val
  int nterm = 0;
  double *rec = nullptr;         // [2][n][4] records {Minv, r, w, s}: launch F_k reads parity (k + 1) & 1, writes parity k & 1 (KB writes parity 1)
};

// Woodbury correction of the Jacobi preconditioner for a FEW dense rows of A (portfolio: k + 1 rows with thousands of entries next to
// n one-entry rows).  With L = the long rows (more than kLongRow entries; at most kWbMaxRows of them),
//     K = K0 + A_L' diag(rho_L) A_L ,   M = D0 + A_L' diag(rho_L) A_L ,   D0 = diag(K0) = diag(P) + sigma + sum_{i not in L} rho_i A_ij^2
//     M^-1 r = y - D0^-1 A_L' S^-1 A_L y ,   y = D0^-1 r ,   S = diag(1 / rho_L) + A_L D0^-1 A_L'          (r_L x r_L, dense, SPD)
// Jacobi alone sees nothing of the rank-r_L block, which carries the large eigenvalues (portfolio: 18.6 PCG iterations per ADMM
// iteration); M is exact whenever K0 is diagonal.  Per application: one pass over the long rows (g = A_L y), an r_L x r_L product
// (h = S^-1 g, S^-1 formed on the host at every rho update: r_L^3 / 3 flops), one pass over their transpose.  Three-kernel PCG form.
constexpr int kWbMaxRows = 128;
// MANY dense rows (lasso: 10 000 sample rows of 5 000 entries next to two-entry rows): the same correction with S formed, factorised
// and inverted on the device -- W = A_L D0^-1/2 as a dense r x ct block (ct = columns the long rows touch), S = W W' + diag(1 / rho_L)
// by one fp64 GEMM, Cholesky + inverse by the ROCm dense solver library (rocBLAS / rocSOLVER, loaded on demand: without them this
// mode is simply off), h = S^-1 g by a dense matrix-vector kernel (r^2 x 8 bytes per application).  Whether M = K (the direct mode) is
// decided NUMERICALLY after every factorisation: M^-1 (K v) must reproduce a probe vector v to 1e-9.
constexpr int kWbLargeMax = 16384;
inline size_t wb_inverse_work(int n) { return (size_t)n * 64 * 3 + 2 * 64 * 64 + 8; }      // dense_hip.hip dense_spd_inverse: column panel, row panel, R, two pivot blocks
// The direct mode in TWO launches per ADMM iteration (wbdirect_hip.hip): P diagonal, every short row of A has exactly one entry, n <= kWbxMaxN.
// A workgroup owns kWbxCols consecutive columns and keeps its dense r x kWbxCols tile of A_L in LDS; the two global reductions of the
// iteration (g = A_L D0^-1 r_0, z~_L = A_L x~) travel as per-workgroup partials, summed in index order by every consumer.
constexpr int kWbxCols = 64;
constexpr int kWbxMaxN = 16384;           // at most 256 workgroups: every consumer folds all partials (G x kWbMaxRows doubles)
struct DevWbx {
  int on = 0, G = 0, nsc = 0;
  int slots = 0;                          // run the two launches as device-scheduled slots (device-driven boundaries; OSQPHipPolicy::device_driven = 2) instead of captured strings of 2 N + 1 launches
  double *tile2 = nullptr;                // [G][kWbMaxRows][kWbxCols] S^-1 A_L by column block (wbx_factor, after every inversion of S)
  double *tile = nullptr;                 // [G][kWbMaxRows][kWbxCols] A_L by column block, zero where A_L has no entry (rows >= r unused)
  double *partG = nullptr, *partZ = nullptr;   // [G][kWbMaxRows] partial sums of the two reductions
  double *ls0 = nullptr, *ls1 = nullptr;  // [3 r] {z, y, z~} of the long rows: read by X from ls0, written by X (workgroup 0) to ls1, handed over by Y
  // ONE launch per ADMM iteration (one = 1; wbdirect_hip.hip k_wbz; round 6).  The second global reduction of the two-launch form is not needed:
  //   z~_L = A_L x~ = A_L x_g + A_L D0^-1 r_0 - (A_L D0^-1 A_L') S^-1 g = A_L x_g + g - (S - diag(1 / rho_L)) h = A_L x_g + h ./ rho_L ,   h = S^-1 g,
  // and A_L x_g is carried along by linearity (Dev::ztg) -- so a launch folds the previous launch's partials of g, forms h with S^-1 held in registers
  // (sinvp: rows padded to kWbMaxRows, loaded at the head of the launch), updates the long rows, its own columns and one-entry rows, builds the next
  // right-hand side and leaves the next partials of g.  The partials alternate between partG and partZ, the long rows' state {z, y, z~, A x_g}
  // between lz0 and lz1, by the parity of the launch (no workgroup reads what its own launch writes).
  int one = 0;
  double *sinvp = nullptr;                // [kWbMaxRows][kWbMaxRows] S^-1, zero-padded (wbx_factor)
  double *lz0 = nullptr, *lz1 = nullptr;  // [4 kWbMaxRows] {z, y, z~, A x_g} of the long rows
  int *sc_ptr = nullptr, *sc_row = nullptr, *sc_src = nullptr;   // per column: its one-entry rows (CSR over columns) and where their values sit in A.val
  double *sc_val = nullptr;
  double *bjj = nullptr;                  // [n] P_jj + sigma (the diagonal of B), refreshed with the tiles
};
struct DevWb {
  int on = 0, r = 0;
  int large = 0;                 // r > kWbMaxRows: device-side dense factorisation; WT unused, W / colmap / ct in use
  int ct = 0;                    // columns with an entry in a long row
  int probe = 0;                 // decide `exact` by the probe after every factorisation (large mode)
  double exact_tol = 1e-6;       // ... the direct mode is on while  max |M^-1 K v - v| <= exact_tol max |v|  (OSQPHipPolicy::woodbury_direct_tol)
  int log = 0;                   // print the checks and the timing of every factorisation (OSQPHipPolicy::woodbury_log)
  int *colmap = nullptr;         // [n] column -> position among the ct touched ones (-1: untouched)
  double *W = nullptr;           // [r][ct] rows of A_L scaled by D0^-1/2 (zero where A_L has no entry)
  double *pv = nullptr;          // [n + m + n] probe vector v, rho .* (A v), v again (the reference M^-1 K v is compared with)
  int *info = nullptr;           // [2] status words of the factorisation / inversion
  int *dbg = nullptr;            // [1] test hook: upcoming device-side inversions to report as failed (OSQPHipPolicy::debug_fail_refactor)
  int exact = 0;                 // K0 is diagonal (P diagonal, every short row of A has one entry): M = K, and M^-1 r_0 IS the solve -- no PCG iteration
                                 // (Engine::run_chunk: KB, the three kernels of M^-1, k_wb_direct, KA); cleared when S^-1 fails its accuracy check
  DevWbx x;                      // exact mode in two launches per ADMM iteration (x.on)
  DevCsr AL, ALT;                // the long rows (r x n) and their transpose (n x r); values gathered from A.val through al_src / alt_src
  int *al_src = nullptr, *alt_src = nullptr;
  unsigned char *islong = nullptr;   // [m]
  int *rows = nullptr;           // [r] row indices in A
  double *WT = nullptr;          // [n][r] dense transpose of the long rows (column j of A_L contiguous): S is formed from it
  double *S = nullptr, *Sinv = nullptr;   // [r][r]
  // large form: the last few inverses, by the rho_bar they were computed for (woodbury_hip.hip wb_factor_large).  A handle that is solved again and
  // again -- parametric re-solves, the steps of a benchmark, which all restart from the setting's rho -- walks the same rho values each time:
  // the r^3 factorisation is then replaced by a look-up + the numerical probe (M^-1 K v = v against the CURRENT matrices and rho vector), which a
  // stale entry cannot pass.  Sinv points at one of the buffers; 8 r^2 bytes each (lasso: 0.8 GB of the 288).
  static constexpr int kCache = 4;
  double rho_key = 0.0;          // rho_bar of the last be::set_rho
  double *cache_buf[kCache] = {nullptr, nullptr, nullptr, nullptr}; double cache_rho[kCache] = {0, 0, 0, 0}; int cache_used = 0, cache_next = 0;
  int cache_hits = 0, cache_on = 1;
  double *g = nullptr, *h = nullptr;      // [r]  (dual form: [max(r, cd)])
  double *Dinv0 = nullptr;       // [n]  1 / D0
  // COLUMN-SPACE ("dual") form of the device-factorised correction (large only; Engine::prepare_wb takes it when it is the smaller system).  The columns
  // the long rows touch split into DENSE columns (two or more long-row entries: cd of them) and SINGLETON columns (exactly one: the lasso's -y_i of
  // the rows  y = A_d x - b).  Eliminating the singletons of every row a (Sherman-Morrison on its block  diag(D0_j) + rho_a s_a s_a') leaves, on the
  // dense columns C,
  //     T = D0_C + A_d' diag(w) A_d ,   w_a = rho_a / (1 + rho_a sigma_a) ,   sigma_a = sum_{j singleton of a} A_aj^2 / D0_j        (cd x cd, SPD)
  // and  M^-1 r  is:  beta_a = sum_{j singleton of a} A_aj r_j / D0_j ;  x_C = T^-1 (r_C - A_d' (w .* beta)) ;  t_a = (beta_a + A_d[a] x_C) / (1 + rho_a sigma_a) ;
  // x_j = (r_j - rho_a A_aj t_a) / D0_j on a singleton column of row a,  x_j = r_j / D0_j on the columns no long row touches.  Exactly the same M as the
  // row-space form (S = 1 / rho_L + A_L D0^-1 A_L', r x r) -- with cd x cd instead of r x r to form, factorise, invert and stream per application:
  // lasso 5k x 10k: cd = 5 000 against r = 10 000 -- an eighth of the factorisation's flops, a quarter of the inverse's bytes.
  // In this form: ct = cd, colmap = index among the dense columns, W [r][cd] = sqrt(w_a) A_d, S = T, Sinv = T^-1 (cd x cd).
  int vendor = 0;                // 1: form / factorise / invert the dense system with rocBLAS + rocSOLVER (A/B switch, OSQPHipPolicy::woodbury_vendor); 0: dense_hip.hip
  double *gjwork = nullptr;      // [wb_inverse_work(order) + 1] panels of the block Gauss-Jordan inverse, the smallest pivot behind them
  int dual = 0, cd = 0;
  int *kind = nullptr;           // [n] 0: no long row touches the column, 1: dense, 2: singleton
  int *dcol = nullptr;           // [cd] the dense columns, ascending
  int *srow = nullptr, *ssrc = nullptr; double *sval = nullptr;      // [n] singleton column -> index a of its long row, position of its entry in A.val, the entry (wb_refresh)
  int *sg_ptr = nullptr, *sg_col = nullptr;                          // per long row: its singleton columns (CSR over the r rows; sums run in list order)
  double *wv = nullptr, *den = nullptr, *beta = nullptr, *wbeta = nullptr, *rt = nullptr;      // [r] w_a, 1 + rho_a sigma_a, beta_a, w_a beta_a, rho_a t_a
  double *uz = nullptr;          // [n] x_C scattered to its columns, ZERO everywhere else (only dense positions are ever written): the vector the row pass gathers
  // FUSED ADMM iteration of the column-space direct mode (fused = 1; woodbury_hip.hip wbf_iteration): the dense block of A is streamed TWICE per ADMM
  // iteration instead of four times (KB's A' v, the g pass, the t pass, KA's A x~).  With c = v - t0 (v = rho z - y, t0 = rho A x_g):
  //   r_0 = sigma x - q - (P + sigma I) x_g + A' c,   so   g_C = [sigma x - q - (P + sigma I) x_g]_C + A_C' (c - w .* beta on the dense rows):
  // ONE transposed pass with the combined vector cc does KB's and the g pass's work; and since x~ = x_g + u,  z~ = A x_g + A u  on a dense row a is
  // ztg_a + (A_d[a] u_C + beta_a - rho_a t_a sigma_a): the t pass already holds A_d[a] u_C, so it also performs KA's z / y update of the dense rows.
  int fused = 0;
  DevCsr Bd, Bn, As;             // views of B / A (same arrays, other block lists): row blocks of B holding a dense column / a column that is not dense; row blocks of A holding a short row
  double *cc = nullptr;          // [m] v - t0 (- w_a beta_a on dense row a)
  int *lidx = nullptr;           // [m] constraint row -> its index among the dense rows (rows[lidx[i]] = i; 0 on short rows)
  double *sig = nullptr;         // [r] sigma_a
  // ... with the dense block held DENSE (dense = 1: cd even, the block at least half full): Ad [r][cd] row-major = A_d, zeros where A has no entry
  // (filled by wb_refresh).  The two passes over the block are then plain dense kernels -- 8 bytes per entry, 16-byte loads, no index stream, no row
  // slices: k_wbf_gd (column sums by row blocks into gp, fixed order) + k_wbf_gr (their sum + the short rows' and P's part from the small lists bq_*)
  // in place of k_wbf_g, and k_wbf_td (two rows per workgroup against ud) in place of k_wbf_t.
  int dense = 0, grb = 0;        // grb: row blocks of the column-sum pass
  int thin = 0;                  // every row of B of a column that is not dense and every short row of A has <= 64 entries: k_wbf_rb / k_wbf_s2 (one thread per row)
  double *Ad = nullptr, *ud = nullptr, *ccd = nullptr, *gp = nullptr;     // ud [cd] = x_C compact (k_wbd_gemv), ccd [r] = cc on the dense rows, gp [grb][cd] partial column sums
  int *bq_ptr = nullptr, *bq_idx = nullptr, *bq_col = nullptr;            // per dense column: its entries of B outside the dense rows (position in B.val, column of B)
};

// indices into Dev::res (results of the residual kernels, reduced on the device)
enum ResId {
  R_PRI_U = 0, R_AX_U, R_Z_U, R_PRI_S, R_AX_S, R_Z_S, R_DY_U, R_DY_S, R_PINF_LHS, R_SUPP,   // m-side
  R_DUA_U, R_PX_U, R_ATY_U, R_DUA_S, R_PX_S, R_ATY_S, R_DX_U, R_DX_S, R_XPX, R_QX, R_QDX,    // n-side
  R_QN_S, R_QN_U,                                                                            // ||q||_inf, ||Dinv q||_inf (scaled q)
  R_ATDY_U, R_ATDY_S, R_PDX_U, R_PDX_S, R_ADX_VIOL,                                          // infeasibility 2nd stage
  R_COUNT
};
// indices into the int32 status block
// F_STAT_STAG: solves that ran into the iteration limit having reduced the residual by less than 10x (a STAGNATING inner solver, as
// opposed to one that converges steadily but needs more iterations than it was given)
enum FlagId { F_DONE = 0, F_ITERS, F_STAT_SUM, F_STAT_MAX, F_STAT_UNCONV, F_STAT_SUMSQ, F_STAT_N, F_STAT_STAG, F_COUNT };
// indices into the fp64 scalar block
enum ScalId { S_TOL_REL = 0, S_TOL_ABS, S_TOL_NOW, S_RN0 /* ||r_0||_inf of the current PCG */, S_RN0H /* the same by parity of the ADMM iteration (slot form): [2] */, S_HIST = 8 /* gamma[kMaxCg+1], alpha[kMaxCg+1], beta[kMaxCg+1] */ };

struct Ctl;       // policy.h: the state block of the chunk-boundary rules (device-driven solves keep one in device memory)
struct Dev {
  int n = 0, m = 0, device = 0;
  Ctl *ctl = nullptr;            // device copy of the driver's state block (nullptr: boundaries are processed by the host)
  double sigma = 0, alpha = 0;
  double rho_eq_factor = 1e3;    // rho_i = rho_eq_factor * rho_bar on equality rows (_osqp.py:27 uses 1e3; see engine.cpp)
  DevCsr A, B;
  int *Bdiag = nullptr;          // position of the diagonal entry of row j inside B.val
  // device-side assembly (setup, update_data_mat): the caller's UNSCALED values in their CSC order and where each goes
  double *Praw = nullptr, *Araw = nullptr;
  int nzP = 0, nzA = 0;
  int *Pi = nullptr, *Pj = nullptr, *Pm1 = nullptr, *Pm2 = nullptr;   // row, column, position in B.val, position of the mirrored entry (-1: diagonal)
  int *Ai = nullptr, *Aj = nullptr, *AmA = nullptr, *AmB = nullptr;   // row, column, position in A.val, position in B.val
  double *cs = nullptr;          // [2] device scalars of the equilibration: cost scale c, this pass's factor
  // problem data (scaled) and scaling
  double *q = nullptr, *l = nullptr, *u = nullptr, *D = nullptr, *Dinv = nullptr, *E = nullptr, *Einv = nullptr;
  double *rho = nullptr, *rho_inv = nullptr;
  int *ctype = nullptr;          // -1 loose, 0 inequality, 1 equality (_osqp.py:516-518)
  // vector updates on the device: the caller's UNSCALED q, l, u stay resident; scaling (_osqp.py:1328, :1357-1358)
  // and the constraint classification (:505-518) are kernels
  double *qraw = nullptr, *lraw = nullptr, *uraw = nullptr;
  int *cnt = nullptr;            // [2] device counters: {inequality rows found by the classification, rows with l > u}
  int eq_from_cnt = 0;           // 1: k_set_rho derives the equality weight from cnt[0] (device classification), 0: rho_eq_factor
  double rho_eq_mixed = 10.0;    // equality weight when inequality rows exist (engine.cpp classify_constraints)
  // ADMM iterates
  double *x = nullptr, *z = nullptr, *y = nullptr, *dx = nullptr, *dy = nullptr;
  double *xs = nullptr;          // x~ : PCG solution, kept as warm start for the next ADMM iteration
  // PCG start of the NEXT ADMM iteration: x~ extrapolated along the last step, xg = x~ + theta (x~ - x~_prev) (KA writes it, KB gathers it
  // and resets xs to it); ztg = A xg by linearity = z~ + theta (z~ - z~_prev), so that t0 = rho .* ztg keeps K xg = B [xg; t0] exact.
  double *xg = nullptr, *xsp = nullptr, *ztg = nullptr;
  int wt = 0;                    // two-kernel form: results written through the XCD's L2 (pcg_hip.hip stw) -- set while matrices + vectors fit the Infinity Cache
  double theta = 0.9;            // extrapolation weight (OSQP_HIP_EXTRAP; 0 = start from the previous x~)
  double *zt = nullptr;          // z~ = A x~
  double *t0 = nullptr;          // rho .* z~  (so K x~ = B [x~; t0] needs no extra SpMV)
  double *v = nullptr;           // rho .* z - y
  // PCG (Chronopoulos-Gear single-reduction form)
  double *r = nullptr, *uu = nullptr, *p = nullptr, *s = nullptr, *w = nullptr, *t = nullptr, *Minv = nullptr;
  double *uu2 = nullptr, *ms = nullptr;  // fused PCG: u_k lives in (k&1 ? uu2 : uu); ms (2n) = interleaved pairs {u_k[j], (Minv .* s_k)[j]}
  int fused = 0;                 // 1: two kernels per PCG iteration (vector update k-1 fused into the SpMV-A kernel of iteration k)
  DevF1 f1;                      // one launch per PCG iteration (slot form only; f1.on)
  DevKf kf;                      // one launch per PCG iteration on the explicit reduced matrix (slot form only; kf.on; never together with f1.on)
  DevWb wb;                      // Woodbury-corrected preconditioner for a few dense rows (three-kernel PCG form; wb.on)
  // reductions
  double *part = nullptr;        // [kPartSlots][kGrid] partial results, slots see backend implementation
  double *res = nullptr;         // [R_COUNT]
  double *scal = nullptr;        // [S_HIST + 2*(kMaxCg+1)]
  int *flags = nullptr;          // [F_COUNT]
  int *slot = nullptr;           // [16] the two phase records of the slot kernels (backend_hip.hip "slot kernels"); nullptr: not used
  void *stream = nullptr;        // hipStream_t (product) / unused (host simulator)
  void *bside = nullptr, *bev0 = nullptr, *bev1 = nullptr;      // batch path: second stream + fork / join events (batch_hip.hip, created on first use)
  void *impl = nullptr;          // backend private (events, graphs, pinned staging)
};

// Batched small-QP solve (batch_hip.hip): nbatch problems sharing the solver's scaled A / B, one workgroup each.
constexpr int kBatchNB = 8;             // pivots per block of the batch kernel's substitutions; the band is stored with kBatchNB zeros of padding per column
constexpr int kBatchDirectMaxBw = 64 - kBatchNB;   // band limit of the batch kernel's direct solve: one wave holds the live window of a block

constexpr int kBatchRec = 12;     // doubles per problem in the result record of the batch kernels (OSQP_HIP_BATCH_REC in include/osqp_hip.h)
struct BatchParams {
  int n, m, nbatch;
  DevCsr A, B;
  const double *D, *Dinv, *E, *Einv;
  double c, cinv, sigma, alpha, rho0, eq_factor, eps_abs, eps_rel, eps_pinf, eps_dinf, cg_frac, rho_tol;
  int max_iter, check, rho_interval, cg_max, unscaled, scaling, precond, rho_is_vec, warm;
  const double *q, *l, *u;      // UNSCALED, [nbatch][n] / [nbatch][m]; nullptr = the shared vector q0 / l0 / u0 for every problem
  const double *q0, *l0, *u0;   // UNSCALED shared vectors [n] / [m]
  double *x, *y;                // in: UNSCALED warm start (if warm), out: UNSCALED solution  [nbatch][n] / [nbatch][m]
  double *rec;                  // [nbatch][kBatchRec]: status, iter, obj, prim_res, dual_res, rho, rho_updates, pcg_iters, status_polish, polish seconds,
                                // rho_estimate (_osqp.py:1275, at the ADMM point), reserved
  int *iters_out = nullptr;     // optional [nbatch]: ADMM iterations of every problem (device; feeds the next call's launch order)
  const int *order = nullptr;   // optional [nbatch]: workgroup w solves problem order[w] (longest-expected first: the batch ends with its slowest
                                // problems otherwise; Engine::batch_solve keeps the order of the previous call's iteration counts)
  double *zs = nullptr;         // optional, SCALED z iterates [nbatch][m]: read as the start when warm (a continued solve keeps its z, _osqp.py:1197-1204),
                                // written at the end (single-QP path: the handle's own d.z)
  int variant = 0;              // 0: automatic choice of the kernel variant; 1-5 force one (OSQPHipPolicy::batch_variant)
  int polish = 0, refine = 0;   // direct variants: polish a SOLVED problem in the kernel (reduced KKT on the active set + refine refinement steps)
  double delta = 1e-6;          // polish regularisation (_osqp.py:1740-1754)
  // Direct linear solve (banded Cholesky of K = P + sigma I + A' diag(rho) A under a bandwidth-reducing symmetric
  // permutation, factor held in LDS): bw < 0 selects the PCG path.  Built by Engine::prepare_batch_direct().
  int bw = -1;                  // half bandwidth of the permuted K (<= kBatchDirectMaxBw)
  double eq_factor_direct = 1e3; // equality-row weight of the direct variant: the reference's 1e3 (_osqp.py:27) -- the value 10 of
                                 // the PCG variants exists only to keep K well conditioned for CG (engine.cpp classify_constraints)
  int nents = 0, ntri = 0;
  const int *perm = nullptr;    // [n] position in the permuted order -> variable
  const int *bp_slot = nullptr; // [nnz(B)] band slot (column * (bw + kBatchNB) + row - column) of each (P + sigma I) entry of B in the permuted lower triangle, else -1
  const int *ke_slot = nullptr, *ke_ptr = nullptr;   // band slots that receive A' rho A terms, and their product ranges
  const int *kp_row = nullptr;  // per product: constraint row i (-> rho_i)
  const double *kp_val = nullptr;                    // per product: A_ia * A_ib (scaled values; refreshed before every batch call)
  const int *tri = nullptr;     // [ntri] (a | b << 8), 1 <= a <= b <= bw: the trailing-update pairs of one elimination step
  // Spectral form of the direct solve (engine.hpp BatchSpectral; n <= kBatchSpecN): V [kBatchSpecN x kBatchSpecN, column-major, zero-padded], lambda [kBatchSpecN],
  // the constraint classes and the equality weight they were built for.  nullptr: not available.  A problem whose own bounds give other classes is
  // left to the banded kernel: the spectral launch marks its record (kBatchUnsolved in rec[0]), a second launch with only_marked solves the marked ones.
  const double *sp_V = nullptr, *sp_lam = nullptr; const int *sp_ctype = nullptr;
  double sp_rho_ref = 0.0, sp_eqf = 0.0;
  // K^-1 for rho_bar = sp_K0_rho (the batch's starting rho: the same for every problem), built on the host with the kernel's own operation order and laid
  // out as the threads hold it -- [64 columns][256 threads]: a problem LOADS its first K^-1 (coalesced, L2-resident) instead of rebuilding it from V
  const double *sp_K0 = nullptr; double sp_K0_rho = 0.0;
  int only_marked = 0;
  // PER-PROBLEM MATRICES (the reference's forward with a P_val / A_val per batch element, /root/reference/src/osqp/nn/torch.py:128-157, 184-217: one
  // solver object per element, each set up -- and therefore SCALED -- with its own matrices).  mat_on != 0: problem b reads its own scaled values
  // Aval_b + b nnz(A), Bval_b + b nnz(B), its own equilibration D_b / Dinv_b (+ b n), E_b / Einv_b (+ b m), c_b[b] and, in the banded direct variant,
  // its own products kp_val_b + b nprod -- all written by be::batch_prepare (k_batch_prepare: assembly + Ruiz equilibration of _osqp.py:389-497 per
  // problem, one workgroup each, in LDS) right before the solve launch.  The spectral form (shared V) does not apply.
  int mat_on = 0, nprod = 0;
  double *Aval_b = nullptr, *Bval_b = nullptr, *D_b = nullptr, *Dinv_b = nullptr, *E_b = nullptr, *Einv_b = nullptr, *c_b = nullptr, *kp_val_b = nullptr;
  const int *kp_a = nullptr, *kp_b = nullptr;   // per product: positions of A_ia, A_ib in A.val (the products are formed per problem)
  // ONE WAVE PER PROBLEM, spectral form (batch_hip.hip k_batch_wave; built by Engine::prepare_batch_wave).  A workgroup of kBatchWaveW waves keeps V and
  // the shared matrices' values in LDS once for all of them; a wave keeps its problem's iterates in registers, rows spread over the lanes: row r of A' (and
  // of every n-vector) belongs to lane r % 64, slot r / 64; the rows of A are sorted by length (descending) first and the sorted position decides
  // lane and slot (wv_row: position -> row, -1 = none), so that the rows one ELL step treats together have similar lengths.  ELL steps of a group of 64 rows:
  // [wv_aend[g - 1], wv_aend[g]); entry (step s, lane l): value A.val[wv_Aidx[s 64 + l]] (index -1: padding, value 0), column wv_Acol[s 64 + l].
  // A' the same over B's entries with column >= n (wv_Tcol = the constraint ROW, original numbering).
  int wv_on = 0;
  int wv_aend[4] = {0, 0, 0, 0}, wv_tend[2] = {0, 0};
  const int *wv_Aidx = nullptr, *wv_Acol = nullptr, *wv_Tidx = nullptr, *wv_Tcol = nullptr, *wv_row = nullptr;
  int wv_first = 0;              // positions [0, wv_first) of the launch order are not the wave kernel's (be::batch_solve sends them to the workgroup kernel)
  int wv_split = 0;              // how many of the first positions the engine wants treated that way (0: none; needs a launch order)
  int wv_cus = 0;                // ... and on how many CUs (the wave kernel's grid leaves them free; wv_split > wv_cus: several problems per CU, one after the other)
  int *wv_queue = nullptr;       // device counter: the next position of the launch order not yet taken by a wave (zeroed before the launch)
};
constexpr int kBatchWaveW = 8;          // waves (= problems in flight) per workgroup of the wave-per-problem kernel; one workgroup per CU
constexpr int kBatchWaveSA = 128, kBatchWaveST = 128;   // most ELL steps of A / A' (what decides is the LDS they take: be::batch_wave_lds_bytes)
constexpr int kBatchSpecN = 128;        // the spectral form keeps K^-1 in registers: row i = thread / 2, 64 columns per thread
constexpr double kBatchUnsolved = -1000.0;

namespace be {

size_t batch_lds_bytes(int n, int m);                       // 0 if a problem does not fit one workgroup's LDS
// stream == nullptr: on d.stream, synchronous.  Otherwise enqueued on that hipStream_t and NOT waited for.  OSQP_FUNC_NOT_IMPLEMENTED if it does not fit
int batch_solve(Dev &d, const BatchParams &p, void *stream = nullptr);
size_t batch_direct_lds_bytes(int n, int m, int nnz, int bw); // 0 if the banded factor does not fit next to the iterates
bool batch_direct_selected(const BatchParams &p);              // would batch_solve run a direct (banded LDL') variant for p?
void batch_release(Dev &d);                                    // the batch path's second stream and events
size_t batch_wave_lds_bytes(int n, int m, int steps);          // LDS of the wave-per-problem kernel (V + `steps` ELL steps of values + one staging vector per wave); 0: does not fit
void batch_products(Dev &d, int nprod, const int *a, const int *b, double *out);   // out[p] = A.val[a[p]] * A.val[b[p]]
// per-problem matrices: Px_b [nbatch][nnz(P as given at setup: upper triangle, CSC order)] / Ax_b [nbatch][nnz(A), CSC order], UNSCALED, device
// pointers; nullptr = the solver's own values for every problem.  Fills p.Aval_b .. p.kp_val_b (allocated by the caller) on `stream` (nullptr: d.stream).
// scaling_iters: the setting's `scaling`.  Returns OSQP_FUNC_NOT_IMPLEMENTED when a problem's matrices do not fit one workgroup's LDS.
int batch_prepare(Dev &d, const BatchParams &p, const double *Px_b, const double *Ax_b, int scaling_iters, void *stream);
void batch_order(Dev &d, int nbatch, const int *iters, int *order, void *stream);  // order = problems by descending iters (ties by index), on `stream`

const char *name();
int init(Dev &d, int device);            // select device, create stream; returns 0 or osqp_error_type
void destroy(Dev &d);
void *alloc(Dev &d, size_t bytes);       // zero-initialised device memory
void dfree(Dev &d, void *p);
void h2d(Dev &d, void *dst, const void *src, size_t bytes);
void d2h(Dev &d, void *dst, const void *src, size_t bytes);   // synchronous w.r.t. the solver stream
void zero(Dev &d, void *dst, size_t bytes);
void sync(Dev &d);
// hipEvent pair on the solver's stream around one osqp_solve ("hipEvent around solve"): mark(0) at its start, mark(1) at its end;
// ev_ms waits for the second event and returns the elapsed milliseconds (host simulator: wall clock)
void ev_mark(Dev &d, int which);
double ev_ms(Dev &d);
void activate(Dev &d);                  // make d.device current for the calling thread (HIP's current device is thread-local)
// Work enqueued on a CALLER's stream that reads solver-owned memory (batch_solve with a stream): ext_record marks its end on that
// stream, ext_wait makes the host wait for it before the solver overwrites or frees that memory (no-op when nothing is pending).
void ext_record(Dev &d, void *stream);
void ext_wait(Dev &d);

// ---- ADMM hot path (all asynchronous on d.stream) ----
// KB: x-part of the rhs and the PCG start, one pass over B (two sums per row):
//   rhs_j = sigma x_j - q_j + (A' v)_j                         (_osqp.py:649-650 folded into the reduced system)
//   r_j   = rhs_j - (B [xs; t0])_j ;  u_j = Minv_j r_j
//   partials: gamma0 = <r,u>, ||r||_inf, ||rhs||_inf ; resets F_DONE/F_ITERS
void kb_rhs(Dev &d);
// K1_i: if PCG already converged -> no-op.  Else test ||r_i||_inf <= max(tol_rel*||rhs||_inf, tol_abs); on success set
//   F_DONE, F_ITERS = i and return; otherwise t = rho .* (A u).
void k1(Dev &d, int i);
// K2_i: if !done: w = B [u; t] ; partial delta = <w,u>
void k2(Dev &d, int i);
// Kv_i: if !done: alpha_i,beta_i from (gamma_i, delta_i, gamma_{i-1}, alpha_{i-1});
//   p = u + beta p ; s = w + beta s ; xs += alpha p ; r -= alpha s ; u = Minv r ; partials gamma_{i+1}, ||r||_inf
void kv(Dev &d, int i);
// KA: after the PCG (budget = number of (K1,K2,Kv) triples that were enqueued):
//   z~ = A xs ; z,y update (_osqp.py:682-703) ; v = rho z - y ; t0 = rho z~ ; dy ; and, on extra workgroups,
//   x = alpha xs + (1-alpha) x ; dx (_osqp.py:660-668).  Also folds the PCG statistics of this ADMM iteration.
void ka(Dev &d, int budget);
// Slot form of a chunk (device-side scheduling of KB / K1 / K2F / K1F / KA; see backend_hip.hip): slot_begin(target, cap) once, then any
// number of slot_pair() launches; slot_done() = ADMM iterations the chunk had completed at the last fetch_flags / fetch_res_flags.
bool slots_supported(const Dev &d);
void slot_begin(Dev &d, int target, int cap);
void slot_pair(Dev &d);
int slot_done(Dev &d);
int slot_seq(Dev &d);                      // slots executed since slot_begin (consistency check of the record hand-over)
// launches one ADMM iteration with `pcg` PCG iterations needs in the slot form (a slot_pair() is two of them):
//   two-kernel form  2 (pcg + 2)   KB, K1, pcg x (K2F, K1F), the K2F that detects convergence, KA
//   F1 form          pcg + 2       F_0, F_1 .. F_pcg, KA (run by the launch whose scalar fold detects convergence; it also leaves the slices F_0 builds
//                                 r_0 from -- no KB launch) + one launch at the start of every chunk
//   Woodbury direct mode in two launches (wbdirect_hip.hip): 2 per iteration + one closing pair per chunk
//   K form           pcg + 3       KB (rhs and r_0 from B), F_0 .. F_pcg on the explicit K, KA (run by the launch whose fold detects convergence)
inline double slot_launches(const Dev &d, double pcg) { return (d.wb.on && d.wb.exact && d.wb.x.on && d.wb.x.slots) ? 2.16 : (d.f1.on ? pcg + 2.0 : (d.kf.on ? pcg + 3.0 : 2.0 * (pcg + 2.0))); }
bool kf_supported();                       // the K form exists (false: the host simulator)
void kf_values(Dev &d, int cond = 0);      // K.val <- term lists with the current rho / A.val / B.val (no-op without the form; cond: inside a boundary group, only when it updated rho)
void f1_refresh(Dev &d);                   // f1.pval <- B.val (no-op without a plan)
bool wb_supported();
bool wbx_supported();                      // the two-launch direct mode exists (false: the host simulator)
void wbx_init(Dev &d);                     // once per handle, after the plan is uploaded (LDS attribute of its kernels on d.device)
void wbx_refresh(Dev &d);                  // tiles / one-entry-row values <- A.val
void wbx_factor(Dev &d, int cond = 0);     // tile2 = S^-1 A_L (after S^-1 has changed); cond: inside a boundary group, only when it updated rho
void wbx_slot_pair(Dev &d);                // the two launches as a pair of slots (device-side scheduling)
inline bool wbx_active(const Dev &d) { return d.wb.on && d.wb.exact && d.wb.x.on; }
inline bool wbx_slots(const Dev &d) { return wbx_active(d) && d.wb.x.slots; }
void wbx_chunk(Dev &d, int niter);         // niter ADMM iterations: X(rhs), { Y, X } x (niter - 1), Y, X(update): 2 niter + 1 launches on d.stream
bool wb_large_supported();                 // the dense solver libraries could be loaded                       // Woodbury preconditioner available (false: the host simulator)
void wb_refresh(Dev &d);                   // wb.AL / ALT / WT values <- A.val (after assembly / equilibration / matrix updates)
void wb_direct(Dev &d);                    // exact mode: x~ = x_g + M^-1 r_0 (after kb_rhs + wb_apply(0)); marks the solve as converged after one step
inline bool wbf_active(const Dev &d) { return d.wb.on && d.wb.exact && d.wb.dual && d.wb.fused; }
void wbf_iteration(Dev &d);                // one ADMM iteration of the fused column-space direct mode: seven launches (no KB, no KA)
void wb_apply(Dev &d, int parity, int direct = 0);   // direct: exact mode -- the last of the three kernels also forms x~ = x_g + u and marks the solve as converged after one step
//         // u = M^-1 r with the partials gamma = <r, u>, ||r||_inf in the slots of `parity` (after kb_rhs: 0, after kv(i): (i + 1) & 1)
// ---- device-driven chunk boundaries (policy.h; backend_hip.hip "boundary kernels").  The host uploads the state block once per solve,
// then only feeds launches: strings of slot launches and, after each chunk's worth, one boundary group -- conditional residual
// kernels, k_decide (the rules of policy.h on the device: termination, rho, tolerance, budget, next chunk), conditional rho update.
bool ctl_supported(const Dev &d);
void ctl_upload(Dev &d, const Ctl &c);     // host -> device copy of the state block (stream-ordered)
void ctl_begin(Dev &d);                    // start the chunk the state block describes (slot record, PCG tolerance, statistics reset)
void ctl_group(Dev &d, int diagonal);      // enqueue one boundary group (diagonal: the Jacobi preconditioner follows rho)
void ctl_poll(Dev &d, Ctl *out, int *seq, int *done);   // snapshot of the state block + slots executed + ADMM iterations of the chunk in flight, read on the side stream WITHOUT waiting for d.stream
void ctl_download(Dev &d, Ctl *out);       // after a stream synchronisation
constexpr int kSlotInts = 24;  // Dev::slot: two phase records of 8 words + the chunk epoch
void slot_poll(Dev &d, int *seq, int *done); // the same two numbers of the RUNNING chunk, read on a side stream without waiting for the launches

// ---- every check_termination iterations ----
// residual norms / objective pieces of (x,z,y) -> d.res[0 .. R_QDX]   (_osqp.py:705-794, 880-908)
void residuals(Dev &d);
void infeas_primal(Dev &d);                         // d.res[R_ATDY_*] = || (Dinv) A' dy ||_inf      (_osqp.py:815-818)
void infeas_dual(Dev &d, double thr, int unscaled);               // d.res[R_PDX_*], d.res[R_ADX_VIOL]              (_osqp.py:846-872)
void fetch_res(Dev &d, double *host_res);           // D2H of d.res + stream sync
void fetch_flags(Dev &d, int *host_flags);          // D2H of d.flags (+ reset of the F_STAT_* counters)
void fetch_res_flags(Dev &d, double *host_res, int *host_flags);   // both behind ONE stream synchronisation (per termination check)

// ---- rho / preconditioner ----
// rho_i by constraint type (_osqp.py:1590-1594), rho_inv, v = rho z - y, t0 = rho z~
void set_rho(Dev &d, double rho_bar);
// Minv_j = 1 / (B_jj + sum_i rho_i A_ij^2)   (diag of K; B_jj already contains sigma).  precond==0 -> Minv = 1
void precond(Dev &d, int diagonal);
void set_pcg_tol(Dev &d, double tol_rel, double tol_abs);

// full != 0 (after warm_start / cold_start, _osqp.py:1493-1509):  z = A x ; xs = x ; dx = dy = 0, then the refresh;
// refresh (always):  zt = A xs ; t0 = rho zt ; v = rho z - y   (state the PCG start of kb_rhs relies on)
void init_iterates(Dev &d, int full);

// tmp = z + y ; z = clip(tmp, l, u) ; y = tmp - z   (_osqp.py:676-680, used by polish :1780)
void project_normalcone(Dev &d);

// Two-kernel PCG iteration available?  When true the driver enqueues  k1(i); k2(i)  per iteration and kv(i) only after the
// last budgeted one: k1(i), i >= 1, performs the vector update of iteration i-1 itself (backend_hip.hip k_k1f).
bool pcg_fused(const Dev &d);

// ---- launch batching ----
bool graphs_supported();
void graph_begin(Dev &d);                 // start capturing d.stream
void *graph_end(Dev &d);                  // stop capturing, instantiate; returns executable-graph handle
void graph_launch(Dev &d, void *g);
void graph_free(Dev &d, void *g);

// ---- assembly / scaling on the device ----
// false: the driver scales on the host and uploads finished value arrays (the test-only host simulator).
bool device_assembly();
// A.val, B.val <- scatter of Araw / Praw through the maps, scaled as  E A D  and  c D P D  when scaled != 0, with sigma on B's
// diagonal when with_sigma != 0   (_osqp.py:432-436, 464, 1443, 1463)
void assemble(Dev &d, int scaled, double c, int with_sigma);
// Ruiz equilibration + cost normalisation (_osqp.py:389-497) of A.val, B.val (assembled unscaled, no sigma) and d.q IN PLACE:
// fills d.D, d.E, d.Dinv, d.Einv, adds sigma to B's diagonal, returns the cost scale c.  iters == 0: D = E = 1, c = 1.
double ruiz(Dev &d, int iters);

// ---- vector updates / warm start on the device ----
bool device_vec_updates();               // false: the host simulator (the driver scales on the host and uploads finished vectors)
// dst (device) <- src, asynchronous on the solver's stream; src_on_device = 0: host memory (reusable when the call returns)
void copy_in(Dev &d, void *dst, const void *src, size_t bytes, int src_on_device);
void stream_wait(Dev &d, void *caller_stream);       // the solver's stream waits for everything queued on caller_stream so far
void gather(Dev &d, double *dst_dev, const double *src_dev, const int *idx_dev, int cnt);   // dst[k] = src[idx[k]] (reordered problems: caller's numbering -> the engine's)
void scale_q(Dev &d, double c);                       // q = c D qraw                                  (_osqp.py:1328)
void scale_bounds(Dev &d, int rho_is_vec);            // l = E lraw, u = E uraw; ctype; cnt[0]       (:1357-1358, :505-518)
int count_bad_bounds(Dev &d, const double *l_dev, const double *u_dev);   // rows with !(l <= u) (:1348-1349); synchronises
void scale_warm(Dev &d, const double *x_dev, const double *y_dev, double c);   // x = Dinv x_in ; y = c Einv y_in  (:1493-1545); NULL: keep

// ---- probes / tests ----
bool ktrace_read(Dev &d, unsigned long long *out, int count);   // diagnostic build only (OSQP_HIP_KTRACE); false otherwise
void test_spmv(Dev &d, int which, const double *in_dev, double *out_dev);   // 0: out = A in ; 1: out = B in
float time_kernel(Dev &d, int which, int reps);                             // mean ms per launch

}  // namespace be
}  // namespace osqp_hip
