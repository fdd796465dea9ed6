// engine.hpp -- host side of the MI355X OSQP engine: problem setup (validation, Ruiz scaling, CSR/B assembly),
// the ADMM driver loop, termination / infeasibility / adaptive-rho logic.  All per-iteration arithmetic runs in
// the backend (backend.h); the host only sees R_COUNT doubles every check_termination iterations.
#pragma once
#include <array>
#include <map>
#include <string>
#include <utility>
#include <vector>

#include "../../include/osqp_hip.h"
#include "backend.h"
#include "policy.h"

namespace osqp_hip {

struct HostCsc {
  int nr = 0, nc = 0;
  std::vector<int> p, i;
  std::vector<double> x;   // UNSCALED values as given by the caller
  int nnz() const { return (int)i.size(); }
};

class Engine {
 public:
  Engine();
  ~Engine();
  // C-API entry points (names follow include/osqp_hip.h)
  int setup(const OSQPCscMatrix *P, const double *q, const OSQPCscMatrix *A, const double *l, const double *u, int m,
            int n, const OSQPSettings *s);
  int solve();
  int solve_impl();
  int warm_start(const double *x, const double *y, bool keep_z = false);
  int cold_start();
  int update_data_vec(const double *q, const double *l, const double *u);
  // the same with DEVICE pointers (this solver's GPU), ordered after the work queued on `stream` so far (NULL: nothing to wait for)
  int update_data_vec_device(const double *q, const double *l, const double *u, void *stream);
  int warm_start_device(const double *x, const double *y, void *stream);
  int update_data_mat(const double *Px, const int *Px_idx, int P_n, const double *Ax, const int *Ax_idx, int A_n);
  int update_settings(const OSQPSettings *s);
  int update_rho(double rho);
  int get_stats(OSQPHipStats *out);
  int set_policy(const OSQPHipPolicy *p);
  int get_policy(OSQPHipPolicy *p) const;
  static void default_policy(OSQPHipPolicy *p);
  static void set_default_policy(const OSQPHipPolicy *p);
  int time_kernel(int which, int reps, double *ms);
  int test_spmv(int which, const double *in, double *out);
  int trace_read(unsigned long long *out, int count);
  int get_scaling(double *D, double *E, double *c);
  int get_reordering(int *perm_cols, int *perm_rows) const;
  // verbose output (include/osqp_hip.h osqp_hip_set_print): one function per handle, the process default at construction
  void set_print(osqp_hip_print_fn fn, void *user) { print_fn_ = fn; print_user_ = user; }
  static void set_default_print(osqp_hip_print_fn fn, void *user);
  int set_rho_eq_factor(double f);
  // Px / Ax: optional PER-PROBLEM matrix values (nbatch x nnz(P upper triangle as given at setup) / nbatch x nnz(A), CSC order; nullptr = this solver's
  // values for every problem) -- the reference's forward with a P_val / A_val per batch element (nn/torch.py:128-157): still ONE launch
  int batch_solve(int nbatch, const double *q, const double *l, const double *u, double *x, double *y, double *rec, int warm, double *zs_dev = nullptr, const double *Px = nullptr, const double *Ax = nullptr);
  int batch_solve_device(int nbatch, const double *q, const double *l, const double *u, double *x, double *y, double *rec, int warm, void *stream, const double *Px = nullptr, const double *Ax = nullptr);
  int attach_batch_matrices(BatchParams &p, const double *Px_dev, const double *Ax_dev, void *stream);      // per-problem matrices: scratch + be::batch_prepare
  void fill_batch_params(BatchParams &p, int nbatch, int warm);
  // LinSysSolver slot (include/osqp_hip.h): this Engine instance is then used ONLY as the reduced-KKT solver
  int ls_setup(const OSQPCscMatrix *P, const OSQPCscMatrix *A, const double *rho_vec, const OSQPSettings *s);
  int ls_set_rho_vec(const double *rho_vec);
  int ls_warm_start(const double *x);
  int ls_solve(double *b, double tol_rel, double tol_abs, int *iters);

  OSQPSolver pub{};          // what the caller holds
  OSQPSettings settings{};
  OSQPInfo info{};
  OSQPSolution solution{};
  int n = 0, m = 0;

  static int validate_settings(const OSQPSettings *s, bool at_setup);

 private:
  osqp_hip_print_fn print_fn_ = nullptr; void *print_user_ = nullptr;
  void say(const char *fmt, ...) const __attribute__((format(printf, 2, 3)));      // one piece of verbose text -> the handle's print function (stdout without one)
  int log_printed_ = 0;                 // entries of Ctl::log already printed in this solve
  void print_summary_line(int iter, double obj, double pri, double dua, double rho, double t0) const;
  void print_log(const Ctl &c, double t0);
  void print_footer() const;
  void print_setup_header() const;
  double rho_at_last_check_ = 0;        // rho_bar the last termination check ran with (the rho column of the last printed line)
  // ---- host copies ----
  HostCsc P_, A_;                       // P_: upper triangle
  std::vector<double> q0_, l0_, u0_;    // unscaled
  std::vector<double> D_, E_, Dinv_, Einv_; double c_ = 1.0, cinv_ = 1.0;
  std::vector<int> ctype_;
  std::vector<double> ls_, us_;         // scaled bounds currently on the device
  std::vector<double> sol_x_, sol_y_, sol_pc_, sol_dc_;
  // maps for value updates
  std::vector<int> Pmap1_, Pmap2_, AmapA_, AmapB_, bdiag_;
  std::vector<double> Aval_, Bval_;     // host mirrors of the scaled device value arrays
  // ---- device ----
  Dev d_;
  bool dev_ready_ = false;
  // ---- driver state ----
  double rho_bar_ = 0.1;
  double eq_factor_mixed_ = 10.0;     // see classify_constraints()
  bool eq_factor_env_ = false;        // policy rho_eq_factor given
  double mixed_eq_factor() const;
  bool eq_factor_set_ = false;        // osqp_hip_set_rho_eq_factor was called: the batch path's direct variant honours it too
  int cg_budget_ = 0;
  bool first_run_ = true;
  bool have_tol_ = false;
  bool use_graph_ = true;
  std::map<std::pair<int, int>, void *> graphs_;
  double *bbuf_ = nullptr; size_t bbuf_cap_ = 0;      // device scratch of batch_solve, kept across calls
  double *bmat_ = nullptr; size_t bmat_cap_ = 0;      // per-problem matrices: scaled values, equilibration and products of every problem (BatchParams::Aval_b ..), kept across calls
  int *d_batch_iters_ = nullptr; int d_batch_iters_n_ = 0;      // device-pointer path: iteration counts of the previous call (its records never reach the host)
  std::vector<int> batch_order_; int *d_batch_order_ = nullptr; size_t batch_order_cap_ = 0;   // problems by descending iteration count of the previous batch call
  double *ckpt_ = nullptr;                            // device copy of (x, x~, z, y) taken before a solve's first chunk (cg cap escalation)
  std::vector<double> ls_rho_;                        // LinSysSolver slot: host copy of rho_vec
  std::vector<int> Arp_, Arj_, Brp_, Bj_;             // host copies of the CSR structure of A and B (symbolic work of the batch path)
  // direct (banded Cholesky) linear solve of the batch path: symbolic data, built on first use
  struct BatchDirect {
    bool tried = false, ok = false;
    int bw = -1, nents = 0, nprod = 0, ntri = 0;
    int bw_symbolic = -1;          // bandwidth found by the ordering, also when the device path does not apply (-2: not analysed)
    int *perm = nullptr, *bp_slot = nullptr, *ke_slot = nullptr, *ke_ptr = nullptr, *kp_row = nullptr, *kp_a = nullptr, *kp_b = nullptr, *tri = nullptr;
    double *kp_val = nullptr;
  } bd_;
  void prepare_batch_direct();
  // one wave per problem (backend.h BatchParams::wv_*, batch_hip.hip k_batch_wave): the ELL row layout of A (rows sorted by length) and A', built once per
  // pattern; the values are read from the device matrices at every launch
  struct BatchWave {
    bool tried = false, ok = false;
    int aend[4] = {0, 0, 0, 0}, tend[2] = {0, 0};
    int *Aidx = nullptr, *Acol = nullptr, *Tidx = nullptr, *Tcol = nullptr, *row = nullptr, *queue = nullptr;
  } bwv_;
  void prepare_batch_wave();
  void free_batch_wave();
  void attach_batch_wave(BatchParams &p, int nbatch);
  int batch_wave_last_ = -1;            // OSQPHipStats::batch_wave_split
  // Spectral form of the batch path's direct solve (batch_hip.hip, SPEC): every problem of a batch shares P, A and the constraint classes, and
  // rho enters K only through ONE scalar -- K(rho) = K_ref + (rho - rho_ref) M1, M1 = A' W A (W: 1 on inequality rows, the equality weight on
  // equality rows).  With K_ref = L L', L^-1 M1 L^-T = Q Lambda Q' and V = L^-T Q:  K(rho)^-1 = V diag(1 / (1 + (rho - rho_ref) lambda)) V'.
  // V and lambda are computed ONCE on the host (dense n <= 128: Cholesky + Jacobi eigenvalue sweeps) and shared by every problem; a workgroup
  // keeps K(rho)^-1 in registers, rebuilds it from V at a rho update (no factorisation) and solves with one dense matrix-vector product.
  struct BatchSpectral {
    bool ok = false; int mat_epoch = -1;
    bool failed = false;             // the decomposition was attempted for exactly this key (mat_epoch, eqf, sigma, rho_is_vec, ctype) and rejected: not retried
    double rho_ref = 0, eqf = 0, sigma = 0; int rho_is_vec = -1;
    std::vector<int> ctype;          // the constraint classes V was built for (a problem of a batch whose own bounds give other classes takes the banded kernel)
    double *V = nullptr, *lam = nullptr; int *d_ctype = nullptr;
    std::vector<double> Vh, lamh;    // host copies (padded, column-major) for K0
    double *K0 = nullptr; double k0_rho = 0; bool k0_ok = false;      // K^-1 at the batch's starting rho (BatchParams::sp_K0)
  } bs_;
  void prepare_batch_k0(double rho0);
  int mat_epoch_ = 0;                // bumped by every change of P / A values
  bool prepare_batch_spectral(double rho_ref, double eqf, bool allow_build);      // true: the form is ready for the current key
  void free_batch_spectral();
  void prepare_wb(const std::vector<int> &Arp, const std::vector<int> &Arj);
  std::vector<int> wb_kind_;            // column kinds of the column-space Woodbury form (backend.h DevWb::kind), kept for the block views built after the upload
  void build_wbf_views(const std::vector<int> &rbA, const std::vector<int> &rbB, const std::vector<int> &Arp, const std::vector<int> &Arj, const std::vector<int> &Brp, const std::vector<int> &Bj);
  struct F1Plan {                       // host image of backend.h DevF1 (plan_f1 builds it without touching the device; upload_f1 commits it)
    bool ok = false; int D = 0, pnnz = 0;
    std::vector<int> blk, prp, pcol, psrc; std::vector<unsigned int> ent; std::vector<unsigned short> cptr;
    int mix = 0; size_t nsp = 0; std::vector<int> fcol, fq, sp_ptr;      // per-block mixing (backend.h DevF1::mix)
  };
  // ---- bandwidth-reducing reordering (Engine::compute_reorder): when the one-launch PCG form does not apply to the matrices as given but
  // does after a symmetric permutation of the variables and a permutation of the constraints, the engine works on the PERMUTED problem
  //   P' = P(pc, pc),  q' = q(pc),  A' = A(pr, pc),  l' = l(pr),  u' = u(pr)
  // and every vector crossing the C API is permuted on the way (solutions, certificates, warm starts, data updates, scaling read-back).
  bool reordered_ = false, no_reorder_ = false;
  std::vector<int> pc_, pr_, ipc_, ipr_;     // pc_[new] = old column, ipc_[old] = new (rows: pr_, ipr_)
  std::vector<int> PvalMap_, AvalMap_;       // caller's position in P.x / A.x -> position in the permuted CSC arrays (osqp_update_data_mat by index)
  int *d_pc_ = nullptr, *d_pr_ = nullptr;    // device copies of pc_, pr_ (device-pointer updates: gathers)
  double reorder_ms_ = 0;
  bool compute_reorder(const std::vector<int> &Arp, const std::vector<int> &Arj, const std::vector<int> &Brp, const std::vector<int> &Bj);
  void apply_reorder();
  void clear_reorder();
  template <class T> std::vector<T> to_internal_n(const T *v) const { std::vector<T> o(n); for (int j = 0; j < n; j++) o[j] = v[pc_[j]]; return o; }
  template <class T> std::vector<T> to_internal_m(const T *v) const { std::vector<T> o(m); for (int i = 0; i < m; i++) o[i] = v[pr_[i]]; return o; }
  bool plan_f1(const std::vector<int> &rb, const std::vector<int> &Arp, const std::vector<int> &Arj, const std::vector<int> &Brp, const std::vector<int> &Bj, F1Plan &pl);
  void upload_f1(const F1Plan &pl);
  void prepare_kf(const std::vector<int> &Arp, const std::vector<int> &Arj, const std::vector<int> &Brp, const std::vector<int> &Bj);      // backend.h DevKf
  bool small_direct_applicable();
  int solve_small_direct(double t0);
  void attach_batch_direct(BatchParams &p, bool spectral);      // spectral: the caller is a batch large enough to pay for the host-side decomposition (or it exists already)
  void free_batch_direct();
  OSQPHipStats stats_{};
  double update_time_acc_ = 0;
  bool clear_update_time_ = false;

  void free_all();
  void compute_scaling(std::vector<double> &Px_s, std::vector<double> &Ax_s, std::vector<double> &q_s);
  void scale_matrix_values(std::vector<double> &Px_s, std::vector<double> &Ax_s) const;
  void classify_constraints(const std::vector<double> &l_s, const std::vector<double> &u_s);
  void upload_bounds_and_types();
  void device_scale_vectors(bool q, bool bounds);
  void ensure_host_vectors();           // host mirrors of q, l, u (unscaled), the scaled bounds and the constraint types, refreshed on demand
  bool raw_stale_ = false, scaled_stale_ = false;
  void upload_q();
  void fill_matrix_values(const std::vector<double> &Px_s, const std::vector<double> &Ax_s);
  void run_chunk(int niter, int budget);
  void run_slots(int begin_target, int pairs, int cap);     // slot form: [slot_begin(begin_target)] + pairs x (B slot, A slot)
  bool use_slots_ = true;
  double slot_pred_[3] = {6.0, 6.0, 14.0};   // PCG iterations per ADMM iteration the slot strings are sized for, per chunk kind: first after an
                                             // adaptation point (its PCGs start from an accurately solved iterate: fewer iterations), ordinary, tight
  std::map<std::array<int, 3>, void *> sgraphs_;
  void admm_core(double t0, double *res);
  OSQPHipPolicy pol_{};                 // this handle's policy (include/osqp_hip.h)
  bool pol_explicit_ = false;           // osqp_hip_set_policy was called: the environment no longer overrides the run-time fields
  Ctl ctl_{};                           // state block of the chunk-boundary rules (policy.h)
  void ctl_setup();
  void apply_rho(double rho);
  void info_from_ctl(double t0);
  void exec_chunk_sync(int cnt, int lim, bool with_res, int kind, double *res, int *flags);
  int run_device_driven(double t0, double *res, int *flags);
  void run_group(int diagonal);
  std::vector<int> feed_hist_;          // slot launches per chunk of the previous device-driven solve (Ctl::hist; index = boundaries processed when the chunk began)
  void polish();
  void apply_scaled_bounds(const std::vector<double> &ls, const std::vector<double> &us);
  void drop_graphs();
  // Captured launches take Dev BY VALUE: its scalar fields (theta, alpha, sigma, the equality-weight rule k_set_rho reads) are frozen into
  // every graph.  sync_graph_scalars() compares them with what the graphs were captured with and drops the graphs when they differ; called
  // wherever launches may be replayed (admm_core, exec_chunk_sync -- hence polish; ls_solve launches eagerly and replays nothing).
  double graph_sig_[7] = {0, 0, 0, 0, 0, 0, 0};
  void sync_graph_scalars();
  int check_termination(const double *res, bool approximate);
  void update_gap_info(const double *res, double t0);
  double gap_time_ = 0;
  double rho_estimate(const double *res) const;
  void store_solution();
  void set_status(int status);
  int auto_rho_interval() const;
};

}  // namespace osqp_hip
