/*
 * osqp_hip.h -- C ABI of libosqp_hip.so, the MI355X-native OSQP ADMM engine.
 *
 * Drop-in boundary: these are the entry points the reference's pybind11 layer
 * binds from the un-vendored osqp C core ("osqp_api_functions.h" / "osqp_api_types.h",
 * /root/reference/src/bindings.cpp.in:9-10).  Each declaration cites the call site it replaces.
 * The struct members are the ones the reference reads or writes through its bindings
 * (bindings.cpp.in:405-447 settings, :473-492 info, :64-105 solution, :41-48 CSC matrix).
 *
 * Plain pointers and sizes only: no torch / pybind / HIP types cross this boundary.
 * All functions return an osqp_error_type value (0 = OSQP_NO_ERROR) unless stated otherwise.
 * A solver handle is not thread-safe; distinct handles may be used from distinct threads
 * (bindings.cpp.in:196-201 releases the GIL around osqp_solve; multithread_test.py:38-53).
 */
#ifndef OSQP_HIP_H
#define OSQP_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

typedef int    OSQPInt;    /* OSQP_USE_LONG OFF  (reference CMakeLists.txt:9, interface.py:153) */
typedef double OSQPFloat;  /* OSQP_USE_FLOAT OFF (interface.py:152)                              */

#define OSQP_INFTY ((OSQPFloat)1e30)   /* bindings.cpp.in:340 */

/* bindings.cpp.in:343-346 */
enum osqp_linsys_solver_type { OSQP_UNKNOWN_SOLVER = 0, OSQP_DIRECT_SOLVER, OSQP_INDIRECT_SOLVER };

/* bindings.cpp.in:349-361 (same member order) */
enum osqp_status_type {
  OSQP_SOLVED = 1, OSQP_SOLVED_INACCURATE, OSQP_PRIMAL_INFEASIBLE, OSQP_PRIMAL_INFEASIBLE_INACCURATE,
  OSQP_DUAL_INFEASIBLE, OSQP_DUAL_INFEASIBLE_INACCURATE, OSQP_MAX_ITER_REACHED, OSQP_TIME_LIMIT_REACHED,
  OSQP_NON_CVX, OSQP_SIGINT, OSQP_UNSOLVED
};

/* bindings.cpp.in:364-375 (same member order) */
enum osqp_error_type {
  OSQP_NO_ERROR = 0, OSQP_DATA_VALIDATION_ERROR, OSQP_SETTINGS_VALIDATION_ERROR, OSQP_LINSYS_SOLVER_INIT_ERROR,
  OSQP_NONCVX_ERROR, OSQP_MEM_ALLOC_ERROR, OSQP_WORKSPACE_NOT_INIT_ERROR, OSQP_ALGEBRA_LOAD_ERROR,
  OSQP_CODEGEN_DEFINES_ERROR, OSQP_DATA_NOT_INITIALIZED, OSQP_FUNC_NOT_IMPLEMENTED
};

/* bindings.cpp.in:378-381 */
enum osqp_precond_type { OSQP_NO_PRECONDITIONER = 0, OSQP_DIAGONAL_PRECONDITIONER };

/* bindings.cpp.in:395-400 */
enum osqp_capabilities_type {
  OSQP_CAPABILITY_DIRECT_SOLVER = 0x01, OSQP_CAPABILITY_INDIRECT_SOLVER = 0x02, OSQP_CAPABILITY_CODEGEN = 0x04,
  OSQP_CAPABILITY_UPDATE_MATRICES = 0x08, OSQP_CAPABILITY_DERIVATIVES = 0x10
};

/* Compressed-sparse-column matrix, borrowed zero-copy from the caller (bindings.cpp.in:41-48). */
typedef struct {
  OSQPInt    m;      /* rows */
  OSQPInt    n;      /* columns */
  OSQPInt   *p;      /* column pointers (n+1) */
  OSQPInt   *i;      /* row indices (nzmax), sorted within a column (interface.py:232-235) */
  OSQPFloat *x;      /* values (nzmax) */
  OSQPInt    nzmax;  /* number of stored entries */
  OSQPInt    nz;     /* -1 for CSC (bindings.cpp.in:48) */
} OSQPCscMatrix;

/* The 29 fields bound at bindings.cpp.in:409-447, in that order. */
typedef struct {
  OSQPInt device;                         /* HIP device ordinal */
  enum osqp_linsys_solver_type linsys_solver;   /* only OSQP_INDIRECT_SOLVER is implemented */
  OSQPInt verbose;
  OSQPInt warm_starting;
  OSQPInt scaling;                        /* Ruiz iterations, 0 = off */
  OSQPInt polishing;                      /* polish by re-running the ADMM on the guessed active set (engine.cpp Engine::polish) */
  OSQPFloat rho;
  OSQPInt   rho_is_vec;
  OSQPFloat sigma;
  OSQPFloat alpha;
  OSQPInt   cg_max_iter;                  /* cap on PCG iterations per ADMM iteration (default 50) */
  OSQPInt   cg_tol_reduction;             /* first-chunk PCG tolerance = ||rhs||_inf / cg_tol_reduction */
  OSQPFloat cg_tol_fraction;              /* PCG tolerance = fraction * (scaled ADMM dual residual), non-increasing */
  enum osqp_precond_type cg_precond;
  OSQPInt   adaptive_rho;
  OSQPInt   adaptive_rho_interval;        /* 0 = automatic (2 * check_termination, or 50) */
  OSQPFloat adaptive_rho_fraction;        /* IGNORED (validated > 0 for API parity): the automatic interval (adaptive_rho_interval = 0) is 2 * check_termination, not a fraction of the setup time */
  OSQPFloat adaptive_rho_tolerance;
  OSQPInt   max_iter;
  OSQPFloat eps_abs;
  OSQPFloat eps_rel;
  OSQPFloat eps_prim_inf;
  OSQPFloat eps_dual_inf;
  OSQPInt   scaled_termination;
  OSQPInt   check_termination;            /* interval; 0 = only at max_iter */
  OSQPInt   check_dualgap;                /* termination additionally requires |duality gap| < eps_abs + eps_rel max(|obj|, |dual obj|); honoured by osqp_solve (small QPs
                                             then take the host-driven loop, not the one-launch kernel); the batch entry points have no gap test and IGNORE it */
  OSQPFloat time_limit;
  OSQPFloat delta;
  OSQPInt   polish_refine_iter;
} OSQPSettings;

/* bindings.cpp.in:473-492 */
typedef struct {
  char      status[32];
  OSQPInt   status_val;
  OSQPInt   status_polish;
  OSQPFloat obj_val;
  OSQPFloat dual_obj_val;
  OSQPFloat prim_res;
  OSQPFloat dual_res;
  OSQPFloat duality_gap;
  OSQPInt   iter;
  OSQPInt   rho_updates;
  OSQPFloat rho_estimate;
  OSQPFloat setup_time;
  OSQPFloat solve_time;
  OSQPFloat update_time;
  OSQPFloat polish_time;
  OSQPFloat run_time;
  OSQPFloat primdual_int;
  OSQPFloat rel_kkt_error;
} OSQPInfo;

/* bindings.cpp.in:64-105: host arrays owned by the solver, refreshed by osqp_solve */
typedef struct {
  OSQPFloat *x;               /* n */
  OSQPFloat *y;               /* m */
  OSQPFloat *prim_inf_cert;   /* m */
  OSQPFloat *dual_inf_cert;   /* n */
} OSQPSolution;

typedef struct OSQPWorkspace_ OSQPWorkspace;   /* opaque */

/* bindings.cpp.in:163-176 dereferences solver->settings / ->solution / ->info */
typedef struct {
  OSQPSettings  *settings;
  OSQPSolution  *solution;
  OSQPInfo      *info;
  OSQPWorkspace *work;
} OSQPSolver;

/* bindings.cpp.in:452-463 (codegen is out of scope: the struct exists so the binding compiles) */
typedef struct {
  OSQPInt embedded_mode, float_type, printing_enable, profiling_enable, interrupt_enable, derivatives_enable;
} OSQPCodegenDefines;

/* ---- core API ---- */
OSQPInt osqp_capabilities(void);                                        /* bindings.cpp.in:402 */
void    osqp_set_default_settings(OSQPSettings *settings);              /* bindings.cpp.in:449 */
const char *osqp_version(void);

/* bindings.cpp.in:153.  P: upper-triangular CSC (interface.py:221-222); A: CSC; l/u clamped to
   +-OSQP_INFTY by the caller (interface.py:237-238).  Inputs are copied; nothing is retained. */
OSQPInt osqp_setup(OSQPSolver **solverp, const OSQPCscMatrix *P, const OSQPFloat *q, const OSQPCscMatrix *A,
                   const OSQPFloat *l, const OSQPFloat *u, OSQPInt m, OSQPInt n, const OSQPSettings *settings);
OSQPInt osqp_solve(OSQPSolver *solver);                                 /* bindings.cpp.in:198 */
OSQPInt osqp_cleanup(OSQPSolver *solver);                               /* bindings.cpp.in:160 */
OSQPInt osqp_warm_start(OSQPSolver *solver, const OSQPFloat *x, const OSQPFloat *y);   /* :193, NULL-able */
OSQPInt osqp_cold_start(OSQPSolver *solver);
OSQPInt osqp_update_data_vec(OSQPSolver *solver, const OSQPFloat *q_new, const OSQPFloat *l_new,
                             const OSQPFloat *u_new);                   /* bindings.cpp.in:237, NULL-able */
/* bindings.cpp.in:280: *_new_idx == NULL means "all entries, in CSC order"; Px refers to the upper triangle */
OSQPInt osqp_update_data_mat(OSQPSolver *solver, const OSQPFloat *Px_new, const OSQPInt *Px_new_idx, OSQPInt P_new_n,
                             const OSQPFloat *Ax_new, const OSQPInt *Ax_new_idx, OSQPInt A_new_n);
OSQPInt osqp_update_settings(OSQPSolver *solver, const OSQPSettings *new_settings);    /* bindings.cpp.in:204 */
OSQPInt osqp_update_rho(OSQPSolver *solver, OSQPFloat rho_new);         /* bindings.cpp.in:213 */
void    osqp_get_dimensions(OSQPSolver *solver, OSQPInt *m, OSQPInt *n);   /* codegen/pywrapper/bindings.cpp.jinja:21 */

/* Out of scope (derivatives / codegen): present so the reference binding links; return OSQP_FUNC_NOT_IMPLEMENTED. */
OSQPInt osqp_adjoint_derivative_compute(OSQPSolver *solver, OSQPFloat *dx, OSQPFloat *dy);            /* :302 */
OSQPInt osqp_adjoint_derivative_get_mat(OSQPSolver *solver, OSQPCscMatrix *dP, OSQPCscMatrix *dA);    /* :310 */
OSQPInt osqp_adjoint_derivative_get_vec(OSQPSolver *solver, OSQPFloat *dq, OSQPFloat *dl, OSQPFloat *du);   /* :318 */
OSQPInt osqp_codegen(OSQPSolver *solver, const char *output_dir, const char *prefix, OSQPCodegenDefines *defines);   /* :322 */
void    osqp_set_default_codegen_defines(OSQPCodegenDefines *defines);  /* bindings.cpp.in:463 */

/* ---- extensions of this engine (no reference analogue) ---- */

/* Engine statistics of the last osqp_solve. */
typedef struct {
  double pcg_iters_total;     /* PCG iterations summed over all ADMM iterations           */
  double pcg_iters_max;       /* largest per-ADMM-iteration PCG count                     */
  double pcg_unconverged;     /* ADMM iterations whose PCG hit the budget                 */
  double kernel_launches;     /* kernels enqueued (graph nodes included)                  */
  double graph_launches;      /* hipGraphLaunch calls                                     */
  double gpu_solve_ms;        /* hipEvent time around the ADMM loop                       */
  double nnzA, nnzB;          /* stored entries of A (CSR) and B = [P+sigma I | A'] (CSR) */
  double pcg_fused;           /* 2: ONE launch per PCG iteration (k_slot1, the F1 form: banded A); 3: ONE launch per PCG iteration on the explicit reduced matrix
                                 (k_slotk, the K form: any sparsity of moderate fill); 1: two kernels (k_k2f, k_k1f); 0: three (k_k1, k_k2, k_kv) */
  double batch_direct_bw;     /* half bandwidth of the reduced KKT matrix under the engine's RCM ordering (batch / small-QP direct
                                 solve); -1 before the first batch or small solve, -2 if the pattern is too dense to analyse */
  double cg_cap_escalations;  /* times the last solve doubled its PCG iteration cap because most solves of a chunk stagnated at it */
  double windowed_blocks;     /* row blocks of A and B whose input-vector window is staged in LDS (16-bit local column indices) */
  double row_blocks;          /* row blocks of A and B in total */
  double slot_topups;         /* last solve: chunks whose string of slot launches ended before the chunk did (more launches followed) */
  double f1_replicas;         /* F1 form: replica vectors of the partial A' t (0: the form does not apply to this problem) */
  double woodbury_rows;       /* dense rows of A treated exactly in the preconditioner (0: plain Jacobi) */
  double woodbury_direct;     /* 1: that preconditioner is K^-1 for the current rho (the linear solves run without PCG iterations); 2: ... and the ADMM
                                 iteration runs as two launches (OSQPHipPolicy::woodbury_fused) */
  double preconditioner;      /* OSQP_HIP_PRECOND_*: what the PCG is preconditioned with right now (below) */
  double woodbury_factorisations, woodbury_factor_ms;   /* last solve: re-factorisations of the Woodbury system at rho updates, and their wall time */
  double reordered;           /* 1: the engine works on a permuted copy of the problem (OSQPHipPolicy::reorder) */
  double reorder_ms;          /* time setup spent looking for the permutation (0: not attempted) */
  double woodbury_cache_hits; /* last solve: rho updates served by an inverse this handle had computed for the same rho_bar before (validated by the
                                 numerical probe against the current matrices) -- woodbury_factorisations counts the others; woodbury_factor_ms covers both */
  double f1_far_columns;      /* F1 form with per-block mixing: far columns (spill slots) over all row blocks of A (0: every block fits its window) */
  double woodbury_dual_cols;  /* device-factorised Woodbury form in column space (OSQPHipPolicy::woodbury_dual): order of its dense system = dense columns (0: row space) */
  double woodbury_fused_iteration; /* 1: the column-space direct mode runs its ADMM iteration fused -- seven launches, the dense block of A streamed twice instead of four times (no KB / KA launch);
                                 2: ... with the block held dense (eight launches: the two passes are dense kernels without an index stream);
                                 3: ... and the short rows / columns by one thread each (k_wbf_rb, k_wbf_s2: seven launches) */
  double woodbury_one_launch; /* 1: the Woodbury direct mode of a few dense rows runs ONE launch per ADMM iteration (k_wbz; woodbury_direct = 2 and OSQPHipPolicy::woodbury_fused = 1) */
  double kform_nnz;           /* K form: stored entries of the explicit reduced matrix K = P + sigma I + A' diag(rho) A (0: the form is not in use) */
  double batch_wave_split;    /* last batch solve: -1 = workgroup-per-problem kernels only; >= 0 = the wave-per-problem kernel ran, with this many of the
                                 longest-expected problems on the workgroup kernel beside it (0: no launch order yet) */
} OSQPHipStats;
/* OSQPHipStats::preconditioner.  `cg_precond = OSQP_DIAGONAL_PRECONDITIONER` (bindings.cpp.in:426, the reference's only preconditioner) selects the
   Jacobi family: plain Jacobi M = diag(K), and -- this engine's addition, on by default, OSQPHipPolicy::woodbury / woodbury_large = 0 switch it
   off -- the same diagonal with the rows of A that hold more than 128 entries treated exactly through the Woodbury identity (1..128 such rows:
   r x r system inverted on the host; up to 16384 rows carrying most of A: formed, factorised and inverted on the device with rocBLAS / rocSOLVER,
   which are loaded on demand).  Which one a handle runs is reported here, never implied. */
enum { OSQP_HIP_PRECOND_NONE = 0, OSQP_HIP_PRECOND_JACOBI = 1, OSQP_HIP_PRECOND_JACOBI_WOODBURY = 2, OSQP_HIP_PRECOND_JACOBI_WOODBURY_DENSE = 3 };
OSQPInt osqp_hip_get_stats(OSQPSolver *solver, OSQPHipStats *out);

/* Re-launch one hot-path kernel `reps` times on the solver's stream with the solver's current device state and
   return its mean duration in milliseconds (hipEvent pair on that stream).  which: 0 = SpMV A (K1),
   1 = SpMV B (K2), 2 = PCG vector update (Kv), 3 = rhs kernel (KB), 4 = A x~ + z,y,x update (KA),
   5 = one whole PCG iteration (K1, K2, Kv in sequence; time per sequence) without the reductions of partials,
   6 = the same with them (what a solve executes); 7 / 8 / 9 = sequence 6 without K2 / K1 / Kv, so that (6) - (7) is
   K2's time INSIDE the sequence, i.e. with the caches in the state a solve leaves them (a same-kernel repeat keeps the
   matrix L2-resident and flatters the kernel).  10 = one FUSED PCG iteration (k_k2f, k_k1f: the default form), 11 / 12 = each
   of the two alone, 13 = the pair with the converged flag set (eager-launch cost of an early-exit pair).
   F1 form (one launch per PCG iteration; each of these times TWO consecutive launches, 0 when the form does not apply to the problem):
   14 = the F-only probe kernel without the scalar fold at the head of the launch, 15 = with it (fixed alpha, beta, no stopping test),
   16 = F launches of the slot kernel itself (phase record, scalars from the fold, a stopping test that never fires, record hand-over):
   what a launch costs inside a solve; 17 = KA of the F1 form (z~ = A x~, z / y / x update, slices of r_0 and rhs), 18 = the first launch of a chunk
   (those slices from the vectors in memory).
   Woodbury direct mode (0 when it is not on): 20 = one ADMM iteration of the two-launch form (k_wbx_y + k_wbx_x; `reps` iterations as one chunk, time per
   iteration), 21 = the kernels of M^-1 = K^-1 of the unfused form (k_wb_p1, k_wb_p2 / k_wb_gemv, k_wb_p3; column space: the five k_wbd_* launches),
   23 = one ADMM iteration of the fused column-space direct mode (seven launches: k_wbf_r, k_wbf_beta, k_wbf_g, k_wbd_gemv, k_wbf_t, k_wbf_x, k_wbf_s).
   The kernels run in a side-effect-free "probe" mode or on saved-and-restored state; solver state is unchanged. */
OSQPInt osqp_hip_time_kernel(OSQPSolver *solver, OSQPInt which, OSQPInt reps, double *mean_ms);

/* Batched solve of `nbatch` QPs that share this solver's P, A, scaling and settings and differ in q / l / u (BASELINE
   configs[4]; semantics of the reference's update-style batching, src/osqp/nn/torch.py:128-164, as ONE kernel launch:
   one workgroup per problem, iterates in LDS).  q: nbatch x n, l/u: nbatch x m, row-major, NULL = the solver's current
   vector for every problem.  x: nbatch x n, y: nbatch x m (in: warm start if warm != 0; out: solution, or the
   infeasibility certificate).  rec: nbatch x OSQP_HIP_BATCH_REC doubles {status_val, iter, obj_val, prim_res, dual_res, rho, rho_updates,
   pcg_iters, status_polish, polish_time, rho_estimate, reserved}.  With the `polishing` setting, every SOLVED problem of a directly-solved batch is polished inside
   the kernel (reduced KKT system of the guessed active set, regularised by `delta`, factorised in LDS, `polish_refine_iter` refinement
   steps: /root/reference/src/osqppurepy/_osqp.py:1710-1828); the PCG variants do not polish (status_polish = 0).  The linear system of each ADMM iteration is solved DIRECTLY (banded LDL' of the reduced KKT matrix in LDS under a
   bandwidth-reducing ordering; pcg_iters = 0, equality weight 1e3 as in the reference) when that band fits next to the iterates
   (<= 144 KB, permuted half bandwidth <= 56, <= 4096 stored entries per matrix), by PCG otherwise.  Returns OSQP_FUNC_NOT_IMPLEMENTED when a problem does not fit
   one workgroup's LDS at all (10n + 8m doubles > 64 KB): callers then loop osqp_update_data_vec + osqp_solve.  A repeated batch of the
   same size is launched in the order of the previous call's iteration counts, longest first (scheduling only; results do not depend on it). */
#define OSQP_HIP_BATCH_REC 12
OSQPInt osqp_hip_batch_solve(OSQPSolver *solver, OSQPInt nbatch, const OSQPFloat *q, const OSQPFloat *l, const OSQPFloat *u,
                             OSQPFloat *x, OSQPFloat *y, OSQPFloat *rec, OSQPInt warm);
/* The same solve with EVERY array in device memory of this solver's device (e.g. torch ROCm tensors through data_ptr(); SURVEY 8f
 * rank 2: no PCIe round trip).  Asynchronous: the kernel is enqueued on `stream` (a hipStream_t; inputs must be ready on it and
 * the outputs are valid once it has drained -- the solver handle must not be used again before that); stream == NULL uses the
 * solver's own stream and waits for it.  l <= u is NOT validated on this path.  nbatch == 0 launches nothing and only answers whether the
 * problem fits the batch kernel (OSQP_NO_ERROR / OSQP_FUNC_NOT_IMPLEMENTED): ranks with an empty share of a sharded batch ask this way. */
OSQPInt osqp_hip_batch_solve_device(OSQPSolver *solver, OSQPInt nbatch, const OSQPFloat *q_dev, const OSQPFloat *l_dev, const OSQPFloat *u_dev,
                                    OSQPFloat *x_dev, OSQPFloat *y_dev, OSQPFloat *rec_dev, OSQPInt warm_start, void *stream);

/* The same batch with PER-PROBLEM MATRIX VALUES: the reference's forward accepts a P_val / A_val per batch element and solves the elements concurrently, one
 * solver object each (/root/reference/src/osqp/nn/torch.py:128-157, 184-217).  Px: nbatch x nnz(P) (the upper triangle in the CSC order given at setup --
 * what osqp_update_data_mat takes as Px with Px_idx = NULL, bindings.cpp.in:240-281), Ax: nbatch x nnz(A) (CSC order); NULL = this solver's own values
 * for every problem.  Still ONE solve launch, one workgroup per problem: a launch in front of it assembles and equilibrates every problem's own matrices
 * (the Ruiz scaling of /root/reference/src/osqppurepy/_osqp.py:389-497 with the problem's own P, q, A -- an element is scaled exactly as a solver set up
 * with its data alone), the solve kernel then reads its problem's values; exact linear solves by the banded LDL' of the problem's own K (iteration
 * counts equal the oracle's per element).  The sparsity pattern, the settings and the launch-order history are the handle's.  _device: every array in
 * device memory, asynchronous on `stream` exactly like osqp_hip_batch_solve_device. */
OSQPInt osqp_hip_batch_solve_mat(OSQPSolver *solver, OSQPInt nbatch, const OSQPFloat *Px, const OSQPFloat *Ax, const OSQPFloat *q, const OSQPFloat *l,
                                 const OSQPFloat *u, OSQPFloat *x, OSQPFloat *y, OSQPFloat *rec, OSQPInt warm);
OSQPInt osqp_hip_batch_solve_mat_device(OSQPSolver *solver, OSQPInt nbatch, const OSQPFloat *Px_dev, const OSQPFloat *Ax_dev, const OSQPFloat *q_dev,
                                        const OSQPFloat *l_dev, const OSQPFloat *u_dev, OSQPFloat *x_dev, OSQPFloat *y_dev, OSQPFloat *rec_dev,
                                        OSQPInt warm_start, void *stream);

/* Parametric re-solve with the new data ALREADY ON THE GPU (the reference's update(q, l, u) + solve() loop,
 * src/osqp/nn/torch.py:136-140, /root/reference/src/osqppurepy/_osqp.py:1312-1367 and :1493-1545 restated as kernels):
 * q_dev / l_dev / u_dev / x_dev / y_dev are UNSCALED float64 arrays in device memory of this solver's device, NULL = unchanged.
 * The solver's stream first waits for everything queued on `stream` so far (NULL: the arrays are ready), copies the vectors
 * device-to-device and rescales / re-classifies them with kernels -- no host round trip except the 4-byte l <= u verdict of
 * osqp_hip_update_data_vec_device (OSQP_DATA_VALIDATION_ERROR, nothing changed).  The host-pointer entry points
 * osqp_update_data_vec / osqp_warm_start run the same kernels behind one H2D copy per vector. */
OSQPInt osqp_hip_update_data_vec_device(OSQPSolver *solver, const OSQPFloat *q_dev, const OSQPFloat *l_dev, const OSQPFloat *u_dev, void *stream);
OSQPInt osqp_hip_warm_start_device(OSQPSolver *solver, const OSQPFloat *x_dev, const OSQPFloat *y_dev, void *stream);


/* ---- LinSysSolver slot (north_star's second boundary) -------------------------------------------------------
 * The reference's C core reaches its KKT solver through a table of function pointers created by an init_linsys_solver_*()
 * call; that header is not in the reference tree (the core is fetched at configure time, CMakeLists.txt:31-37), so the
 * member list below follows SURVEY 8b's reconstruction [UPSTREAM-UNVERIFIED]: type, name, solve, update_settings,
 * warm_start, adjoint_derivative, free, update_matrices, update_rho_vec, nthreads.  What is pinned by the reference is the
 * MEANING of solve (purepy's linsys_solver.solve, _osqp.py:307-311, and update_xz_tilde, :644-658):
 *     b = [rhs_x (n); rhs_z (m)]   ->   b = [x~ ; z~],   [[P + sigma I, A'], [A, -diag(1/rho)]] [x~; nu] = b,  z~ = rhs_z + nu / rho
 * which this (indirect) solver computes on the device as  (P + sigma I + A' diag(rho) A) x~ = rhs_x + A'(rho .* rhs_z)  by PCG
 * (warm-started from the previous x~) and  z~ = A x~.  P (upper triangle) and A arrive ALREADY SCALED, rho_vec is given per
 * constraint; the object owns device copies.  b is a host buffer of n + m doubles. */
typedef struct OSQPHipLinSysSolver_ OSQPHipLinSysSolver;
struct OSQPHipLinSysSolver_ {
  enum osqp_linsys_solver_type type;                                              /* OSQP_INDIRECT_SOLVER */
  const char *(*name)(OSQPHipLinSysSolver *self);
  OSQPInt (*solve)(OSQPHipLinSysSolver *self, OSQPFloat *b, OSQPInt admm_iter);    /* 0, or an osqp_error_type */
  void (*update_settings)(OSQPHipLinSysSolver *self, const OSQPSettings *settings);/* cg_max_iter, cg_tol_*, cg_precond */
  void (*warm_start)(OSQPHipLinSysSolver *self, const OSQPFloat *x);               /* start vector of the next PCG (n) */
  OSQPInt (*adjoint_derivative)(OSQPHipLinSysSolver *self);                        /* OSQP_FUNC_NOT_IMPLEMENTED */
  void (*free)(OSQPHipLinSysSolver *self);
  OSQPInt (*update_matrices)(OSQPHipLinSysSolver *self, const OSQPCscMatrix *P, const OSQPInt *Px_new_idx, OSQPInt P_new_n,
                             const OSQPCscMatrix *A, const OSQPInt *Ax_new_idx, OSQPInt A_new_n);   /* same pattern, new values */
  OSQPInt (*update_rho_vec)(OSQPHipLinSysSolver *self, const OSQPFloat *rho_vec, OSQPFloat rho_sc);
  OSQPInt nthreads;                                                                /* 1 host thread drives the device */
  OSQPInt pcg_iters;                                                               /* PCG iterations of the last solve */
  void *impl;
};
/* scaled_prim_res / scaled_dual_res: optional pointers to the caller's CURRENT scaled ADMM residuals; when given, a solve stops
 * at ||r||_inf <= cg_tol_fraction * (*scaled_dual_res) (this engine's rule, DESIGN.md "PCG tolerance"), otherwise (and while
 * that value is not positive) at a relative reduction of 1e-7.  polishing != 0: always the tight relative rule. */
OSQPInt osqp_hip_linsys_init(OSQPHipLinSysSolver **self, const OSQPCscMatrix *P, const OSQPCscMatrix *A, const OSQPFloat *rho_vec,
                             const OSQPSettings *settings, const OSQPFloat *scaled_prim_res, const OSQPFloat *scaled_dual_res,
                             OSQPInt polishing);

/* ---- engine policy (no reference analogue) ----------------------------------------------------------------------------------
 * Everything this engine decides beyond OSQPSettings, PER SOLVER HANDLE: which kernel forms it may use, the rules it adds to the
 * reference's ADMM loop on the indirect path (DESIGN.md sections 2, 2.1, 2.2) and how it schedules its launches.  Defaults =
 * osqp_hip_default_policy(); a new handle starts from the defaults overridden by OSQP_HIP_* environment variables (experiments and
 * A/B runs; read in ONE place, Engine's policy_from_env(), when the handle is created and -- the run-time fields -- at every solve
 * until osqp_hip_set_policy() has been called on it).  osqp_hip_set_policy() changes a handle's policy; fields marked [setup] only
 * matter to handles created afterwards through osqp_hip_set_default_policy() (process-wide default for the NEXT osqp_setup calls of
 * the calling thread), because they choose data structures built at setup. */
typedef struct {
  /* kernel forms */
  OSQPInt graph;              /* replay captured launch strings (hipGraph) instead of enqueuing them one by one                 [setup] */
  OSQPInt slots;              /* device-side scheduling of the ADMM / PCG phases ("slot" kernels)                               [setup] */
  OSQPInt pcg_fused;          /* 1: vector update fused into the SpMV kernels; 0: the three-kernel PCG iteration                [setup] */
  OSQPInt f1;                 /* one launch per PCG iteration where the matrices allow it (banded A; 1: a block's few columns outside its
                                 window are taken as far columns -- band + long-range couplings; 2: strict windows only;
                                 3: diagnostic -- the mixing kernels also on a matrix without far columns)                          [setup] */
  OSQPInt window;             /* windowed row blocks (16-bit local column indices, input window in LDS)                         [setup] */
  OSQPInt woodbury;           /* a few dense rows of A (1..128 rows with > 128 entries) are treated exactly in the preconditioner   [setup] */
  OSQPInt woodbury_direct;    /* ... and when the rest of K is diagonal, that preconditioner IS K^-1: the linear solve without PCG iterations [setup] */
  OSQPInt woodbury_large;     /* up to 16384 dense rows carrying most of A: the same correction with the r x r system formed, factorised and inverted on
                                 the device (rocBLAS / rocSOLVER, loaded on demand; off where they are missing)                    [setup] */
  OSQPInt device_driven;      /* chunk boundaries (termination test, adaptive rho, PCG tolerance / budget) decided on the device; 2: also for the Woodbury
                                 direct mode in two launches (there the host-synchronous loop is the faster one and the default) */
  OSQPInt small_direct;       /* small QPs: the whole solve as ONE launch of the batch kernel's direct (banded LDL') variant */
  OSQPInt batch_reorder;      /* batch solves: launch the problems in the order of the previous call's iteration counts */
  OSQPInt batch_variant;      /* 0 automatic; 1 direct (one wave), 2 direct256, 3 w64 (PCG), 4 w256 (PCG), 5 generic -- if applicable */
  /* rules on top of the reference's loop */
  OSQPFloat extrap;           /* PCG start = x~ + extrap * (x~ - x~_prev)  (0: the previous x~)                                  [setup] */
  OSQPFloat rho_eq_factor;    /* equality-row weight on problems with inequality rows; 0 = automatic (10, or 1e3 for n <= 256)   [setup] */
  OSQPInt rho_window;         /* ADMM iterations before an adaptation point that run with a tighter PCG tolerance (0: none) */
  OSQPFloat rho_window_tol;   /* ... tolerance factor of that window */
  OSQPInt rho_persist;        /* apply a rho estimate that stays on one side of rho at two consecutive adaptation points */
  OSQPFloat rho_tol_exp;      /* adaptive_rho_tolerance is spent as tolerance^rho_tol_exp on QPs (1 = the setting's literal value) */
  OSQPFloat budget_tolerate, budget_sigma; OSQPInt budget_slack, budget_full;   /* PCG limit per solve: mean + sigma * std of the last chunk (+ slack); full: never below cg_max_iter */
  OSQPInt cg_escalate;        /* double cg_max_iter while the inner solver stagnates; checkpointed first chunk */
  OSQPInt stall;              /* drop the PCG tolerance while the iterates run away (unbounded problems) */
  OSQPFloat polish_delta_floor; /* polish on the PCG path: the refinement recurrence runs with delta_eff = max(settings.delta, this) (Engine::polish) */
  OSQPFloat polish_pcg_tol;   /* ... relative residual its inner systems are solved to */
  /* scheduling of the launch strings (results never depend on these) */
  OSQPInt slot_poll; OSQPInt poll_low; OSQPFloat poll_first, poll_frac, poll_wait;      /* host-synchronous chunks: top-ups from polled progress */
  OSQPInt finish_pairs; OSQPInt poll_sleep_us;      /* device-driven chunks: a boundary group goes out when at most finish_pairs slot pairs are missing; pause between polls */
  /* diagnostics */
  OSQPInt slot_log, setup_timing, batch_timing, woodbury_log;
  OSQPFloat woodbury_direct_tol; /* device-factorised form: the direct mode (no confirming PCG iteration) is taken while the probe  max |M^-1 K v - v| / max |v|
                                 measured after every factorisation stays below this (default 1e-6: every linear solve then reduces its residual a
                                 millionfold -- orders beyond any tolerance the PCG would have been asked for; r03 demanded 1e-9, which the Woodbury
                                 identity misses at 5k x 10k by a factor 4-40)                                                        [setup] */
  OSQPInt woodbury_fused;     /* 1 (default): the Woodbury direct mode in its fused forms where they apply -- a few dense rows (P diagonal, one-entry short rows,
                                 n <= 16384; wbdirect_hip.hip): ONE launch per ADMM iteration (round 6; rounds 3-5: two); many dense rows in column space: seven
                                 launches that stream the dense block twice (woodbury_hip.hip wbf_iteration).  2: the two-launch form of rounds 3-5 (A/B runs);
                                 0: KB, the kernels of M^-1, KA                                                                                    [setup] */
  OSQPInt debug_fail_refactor; /* TEST HOOK: that many of the next device-side inversions of the Woodbury system report "inaccurate" (exercises the
                                 hand-over of a device-driven direct-mode solve to the host); 0 in production */
  OSQPInt reorder;            /* 1 (default): when the one-launch PCG form does not apply to the matrices as numbered by the caller, look for a
                                 bandwidth-reducing permutation of variables and constraints under which it does, and work on the permuted problem
                                 (every vector crossing this API keeps the caller's numbering); 0: never; 2: always permute (tests)      [setup] */
  OSQPInt woodbury_cache;     /* 1 (default): the device-factorised Woodbury form keeps the last four inverses by rho_bar; a rho the handle has seen before is a
                                 look-up + the numerical probe instead of an r^3 factorisation (OSQPHipStats::woodbury_cache_hits); 0: always factorise   [setup] */
  OSQPInt kform;              /* 1: where the one-launch form on A alone (f1) does not apply and no Woodbury mode is on, hold the reduced matrix
                                 K = P + sigma I + A' diag(rho) A explicitly when its fill is moderate (sum of squared row lengths of A <= 8 nnz(A)) and run
                                 ONE launch per PCG iteration on it (any sparsity pattern; k + 3 launches per ADMM iteration instead of 2 k + 4).
                                 0 (default): the two-kernel form -- measured faster on MI355X: K needs nnz(K) random gathers per product (41 per row at
                                 configs[1] sizes with unstructured columns: 4.1 M against the 2.2 M of the A / B pair), and a random 8..32-byte gather costs a
                                 whole 128-byte line of L2 -> L1 traffic: 18-24 us per 4.2 M gathers alone (profiles/r06a_kform_gather_bench.txt)        [setup] */
  OSQPInt woodbury_dual;      /* 1 (default): the device-factorised Woodbury form works in COLUMN space where that is the smaller system -- the columns the dense rows touch
                                 split into dense columns (two or more entries) and singletons (one: the lasso's -y_i); with fewer dense columns than 3/4 of the dense
                                 rows, the cd x cd system  T = D0_C + A_d' diag(w) A_d  on the dense columns replaces the r x r system S (lasso 5k x 10k: 5 000 against
                                 10 000: an eighth of the factorisation, a quarter of the inverse); 0: always the row-space form                                [setup] */
  OSQPInt woodbury_vendor;    /* 0 (default): the dense system of the device-factorised Woodbury form is formed and inverted by this engine's own kernels on the fp64 matrix
                                 cores (dense_hip.hip: a strided MFMA GEMM + block Gauss-Jordan inversion); 1: rocBLAS dgemm + rocSOLVER dpotrf / dpotri, loaded on
                                 demand (the route of rounds 3-5, kept for A/B runs; without the libraries osqp_setup falls back to plain Jacobi)            [setup] */
  OSQPInt batch_wave;         /* batch solves in the spectral form: one WAVE per problem, eight problems in flight per CU, the longest-expected problems on the
                                 workgroup kernel beside it (batch_hip.hip k_batch_wave).  0 (default): for batches of at least 2048 problems, or 1280 with a
                                 launch order from a previous call (measured, MPC batch, with a launch order: 4096 QPs 2.8 ms against 5.7, 2048 1.8 against
                                 3.0, 1280 1.7 against 1.9, 1024 1.7 against 1.6; without: 2048 3.5 against 3.6, 1536 3.5 against 3.1 -- below that a
                                 problem's own latency on one wave decides); 1: at every batch size; -1: never */
} OSQPHipPolicy;
void    osqp_hip_default_policy(OSQPHipPolicy *policy);
OSQPInt osqp_hip_set_policy(OSQPSolver *solver, const OSQPHipPolicy *policy);
OSQPInt osqp_hip_get_policy(OSQPSolver *solver, OSQPHipPolicy *policy);
void    osqp_hip_set_default_policy(const OSQPHipPolicy *policy);      /* NULL: back to osqp_hip_default_policy() + environment */

/* Diagnostic builds only (make TRACE=1: kernels stamp the wall clock per workgroup and phase, 16 slots per workgroup):
 * copies the stamps of the most recent launches.  The product library returns OSQP_FUNC_NOT_IMPLEMENTED. */
OSQPInt osqp_hip_trace_read(OSQPSolver *solver, unsigned long long *out, OSQPInt count);
/* Test hook (used by tests/ only): which = 0: out = A in (n -> m); 1: out = B [in_n; in_m] (n + m -> n), with the scaled device matrices. */
OSQPInt osqp_hip_test_spmv(OSQPSolver *solver, OSQPInt which, const OSQPFloat *in, OSQPFloat *out);
/* Weight of equality rows relative to inequality rows, rho_eq = factor * rho, used when equality and inequality rows are
   mixed (default 10; the reference's 1e3 is kept when every active row is an equality).  See engine.cpp (Engine::classify_constraints)
   classify_constraints() for the rationale.  Takes effect immediately (rho vector + preconditioner are rebuilt). */
OSQPInt osqp_hip_set_rho_eq_factor(OSQPSolver *solver, OSQPFloat factor);
/* copy out internal scaling D (n), E (m), c */
OSQPInt osqp_hip_get_scaling(OSQPSolver *solver, OSQPFloat *D, OSQPFloat *E, OSQPFloat *c);
/* the permutation this handle works under (OSQPHipPolicy::reorder): perm_cols[k] (n) / perm_rows[k] (m) = the caller's index of the engine's
   k-th variable / constraint; the identity when the problem was not reordered (OSQPHipStats::reordered = 0).  Diagnostic: nothing at this
   API is expressed in the engine's numbering. */
OSQPInt osqp_hip_get_reordering(OSQPSolver *solver, OSQPInt *perm_cols, OSQPInt *perm_rows);
/* Where `verbose` output goes.  The reference's C core prints through c_print, which its Python binding maps to PySys_WriteStdout under the GIL
   (/root/reference/cmake/printing.h:2-7) while solve() runs with the GIL released (bindings.cpp.in:197-199).  Here: every line of text a handle
   prints is handed to ITS print function, or to stdout (fputs) when none is set.  osqp_hip_set_default_print installs the function handles
   created AFTERWARDS start with (osqp_setup prints the problem header before the caller holds the handle) -- the Python layer installs one that
   writes to sys.stdout (a ctypes callback takes the GIL itself); osqp_hip_set_print changes one handle's.  fn == NULL restores stdout. */
typedef void (*osqp_hip_print_fn)(const char *text, void *user);
void    osqp_hip_set_default_print(osqp_hip_print_fn fn, void *user);
OSQPInt osqp_hip_set_print(OSQPSolver *solver, osqp_hip_print_fn fn, void *user);
/* name of the compute backend compiled into this library: "hip-gfx950" for the product */
const char *osqp_hip_backend(void);

#ifdef __cplusplus
}
#endif
#endif /* OSQP_HIP_H */
