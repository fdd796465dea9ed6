#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X OSQP engine (contract: see the task statement / DESIGN.md §Measurement).

Metric (BASELINE.json): ADMM iterations/sec (+ achieved HBM GB/s of the PCG SpMV) on the n=100k, m=200k, nnz(A)=1M,
nnz(P)=200k sparse QP (BASELINE configs[1], generator problems.banded_qp), indirect PCG.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  N>1:  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one complete cold-started solve of the QP (setup -- scaling, CSR/B assembly, H2D -- is done once, outside the
timed region: all inputs are resident in HBM when the clock starts).  value = ADMM iterations executed by all ranks in the
K timed steps / wall time (max over ranks).  A single QP does not shard (an n-vector all-reduce per PCG iteration would be
latency-bound over xGMI: DESIGN.md §6), so with N > 1 every rank solves its own replica of the same QP on its own GPU -- one
problem per GPU, no data-path collective; the only communication is the final all_gather of {status, iter, obj, prim_res,
dual_res} over RCCL (scaling: weak).  The batched-QP path that really shards a workload is bench_batch.py.

Rank 0 also reports
  roofline      algorithmic bytes per launch / mean launch time (hipEvent pair on the solver's stream) of the dominant PCG
                kernel (the SpMV over B = [P+sigma I | A']), against the 8 TB/s HBM peak;
  cpu_baseline  the oracle (oracle/osqp_oracle.c: the CPU restatement of the reference algorithm with a direct LDL' KKT
                solve -- the role QDLDL plays in the reference's builtin algebra), timed on this host's cores on a bounded
                sample of the same workload.
"""
import argparse
import json
import os
import sys
import time
import warnings

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, 'osqp-python_amd'), os.path.join(ROOT, 'oracle')):
    if _p not in sys.path:
        sys.path.insert(0, _p)

HBM_PEAK_GBS = 8000.0     # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def median(v):
    v = sorted(v)
    return None if not v else (v[len(v) // 2] if len(v) % 2 else 0.5 * (v[len(v) // 2 - 1] + v[len(v) // 2]))


def spmv_bytes(nnz, rows, cols):
    """Algorithmic bytes of one CSR SpMV: fp64 values + int32 column indices, row pointers, the input
    vector gathered once, the output written once."""
    return 12 * nnz + 4 * (rows + 1) + 8 * cols + 8 * rows


def device_state(local):
    """What the bench saw of the GPU right behind its timed region (box-to-box drift of a few per cent shows up in the launch times: round-5 verdict, item 7):
    the device's name / CU count / nominal clock from the runtime, and the current clocks, performance level and power from rocm-smi (None where the tool is
    absent or says nothing parseable).  Reporting only: nothing is set."""
    import shutil
    import subprocess
    out = {}
    try:
        import torch
        p = torch.cuda.get_device_properties(local)
        out.update({'name': p.name, 'cus': int(p.multi_processor_count), 'nominal_clock_mhz': getattr(p, 'clock_rate', 0) / 1e3 or None, 'hbm_gib': round(p.total_memory / 2**30, 1)})
    except Exception as e:          # noqa: BLE001
        out['error'] = repr(e)
    smi = shutil.which('rocm-smi') or '/opt/rocm/bin/rocm-smi'
    try:
        r = subprocess.run([smi, '-d', str(local), '--showclocks', '--showperflevel', '--showpower', '--json'], capture_output=True, text=True, timeout=20)
        js = json.loads(r.stdout[r.stdout.index('{'):]) if '{' in r.stdout else {}
        card = next(iter(js.values())) if js else {}
        out['rocm_smi'] = {k: v for k, v in card.items() if any(t in k.lower() for t in ('sclk', 'mclk', 'fclk', 'socclk', 'performance', 'power'))} or None
    except Exception as e:          # noqa: BLE001
        out['rocm_smi'] = None; out['rocm_smi_error'] = repr(e)[:120]
    return out


def pmc_sources(workload):
    """The kernel sources a PMC summary of `workload` describes (profiles/summarize_pmc.py stamps their hash, pmc_traffic compares it): the PCG kernels and
    the shared helpers always; the Woodbury forms' files for the configurations whose solves launch them."""
    files = ['pcg_hip.hip', 'hip_common.h']
    if workload.startswith('lasso'):
        files.append('woodbury_hip.hip')
    if workload.startswith('portfolio'):
        files.append('wbdirect_hip.hip')
    return files


def pmc_traffic(kernel, workload):
    """HBM-side bytes per launch of `kernel` from the committed rocprofv3 PMC summaries of THIS workload (profiles/*_pmc_<workload>_
    FETCH_SIZE.csv and ..._WRITE_SIZE.csv, produced by profiles/run_pmc.sh with one pass per counter -- PMC passes cannot run inside
    the timed bench): mean over the ACTIVE dispatches; FETCH_SIZE / WRITE_SIZE are in KB, and on gfx950 FETCH_SIZE counts 64 B per
    128-B request, i.e. half the bytes of a coalesced stream (MI355X_MICROARCH.md, HBM section) -> doubled.  None when no summary of
    this workload (same generator, same size) is present -- a figure measured on another problem size is not reported."""
    import csv
    import glob
    import hashlib
    if '+' in kernel:                                   # a launch group (the lasso's fused iteration): the sum over its kernels, None if any is missing
        parts = [pmc_traffic(k, workload) for k in kernel.split('+')]
        return None if any(p is None for p in parts) else float(sum(parts))
    h = hashlib.sha256()
    for f in pmc_sources(workload):
        with open(os.path.join(ROOT, 'osqp-python_amd', 'csrc', f), 'rb') as fh:
            h.update(fh.read())
    sha = h.hexdigest()[:16]
    vals = {}
    for ctr in ('FETCH_SIZE', 'WRITE_SIZE'):
        files = sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_pmc_%s_%s.csv' % (workload, ctr))))
        if not files:
            return None
        rows = list(csv.DictReader(open(files[-1])))
        # a summary is used only if it was taken of THESE kernel sources (profiles/summarize_pmc.py stamps the hash of pmc_sources(workload) as its
        # last row): counters of an earlier state of the kernel are not reported as this run's traffic
        if not any(r['kernel'] == '__source__' and r['counter'] == sha for r in rows):
            return None
        # (templated kernels appear with their arguments -- k_f1_probe<4> --, some rocprofv3 builds add the parameter list: match the bare name)
        for row in rows:
            bare = row['kernel'].split('(')[0].split('<')[0].split(' ')[-1].split('::')[-1]
            if bare == kernel and row['counter'] == ctr and float(row['active_dispatches']) > 0:
                vals[ctr] = float(row['mean_active'])
    if len(vals) != 2:
        return None
    return (2.0 * vals['FETCH_SIZE'] + vals['WRITE_SIZE']) * 1024.0


def _cpu_child(conn, P, q, A, l, u, settings, seconds_target, linsys):
    try:
        conn.send(_cpu_baseline(P, q, A, l, u, settings, seconds_target, linsys))
    except Exception as e:          # noqa: BLE001 -- reported to the parent, which falls back
        conn.send({'error': repr(e)})


def cpu_baseline(P, q, A, l, u, settings, seconds_target=40.0):
    """The oracle in a child process with a hard time limit (3x the budget): the direct LDL' path first; when its ordering +
    factorisation does not fit (dense data blocks: lasso), the oracle's reduced-KKT PCG path on a bounded sample instead -- said
    so in `sample`."""
    import multiprocessing as mp
    ctx = mp.get_context('fork')
    for linsys in (0, 1):
        parent, child = ctx.Pipe(duplex=False)
        pr = ctx.Process(target=_cpu_child, args=(child, P, q, A, l, u, settings, seconds_target, linsys))
        pr.start()
        out = parent.recv() if parent.poll(3.0 * seconds_target + 20.0) else None
        if out is None:
            pr.terminate()
        pr.join(5.0)
        if out is not None and 'error' not in out:
            return out
    return {'value': None, 'unit': 'ADMM iter/s', 'cores': 1, 'kind': 'port', 'sample': 'the oracle did not finish a sample within %.0f s' % (3 * seconds_target)}


def _cpu_baseline(P, q, A, l, u, settings, seconds_target, linsys):
    """Oracle (direct LDL', AMD ordering, 1 thread) on the same QP, cold-started, for at most ~seconds_target of ADMM
    iterations: run to convergence when that fits (then its iteration count and time-to-solution are reported), else a bounded
    sample of iterations."""
    import oracle
    from oracle import Oracle, SOLVED
    oracle.use_native()                          # -O3 -march=native, compiled on THIS host; this is the timing child process
    t0 = time.time()
    ncal = 20 if linsys == 0 else 2              # (the PCG fallback of a dense-block problem takes seconds per ADMM iteration on one core)
    o = Oracle().setup(P, q, A, l, u, eps_abs=settings['eps_abs'], eps_rel=settings['eps_rel'], max_iter=ncal,
                       adaptive_rho_interval=settings['adaptive_rho_interval'], check_termination=settings['check_termination'],
                       linsys=linsys, **({'pcg_max_iter': 200, 'pcg_tol': 1e-7} if linsys else {}))
    t_setup = time.time() - t0
    _, _, info = o.solve()                       # 20 (2) iterations: calibrates the per-iteration cost
    per_it = info.solve_time / max(info.iter, 1)
    k = int(max(ncal, min(20000, seconds_target / max(per_it, 1e-9))))
    o.update_settings(max_iter=k, warm_start=0)
    _, _, info = o.solve()
    done = info.status_val == SOLVED
    out = {'value': info.iter / info.solve_time, 'unit': 'ADMM iter/s', 'cores': 1, 'kind': 'port', 'build': 'gcc -O3 -march=native -fno-fast-math, compiled on this host',
           'sample': '%d cold-started ADMM iterations of the same QP in %.1f s (%s); %s; setup %.1f s not included'
                     % (info.iter, info.solve_time, 'run to convergence' if done else 'bounded sample, not converged',
                        ('direct LDL\' KKT solve, own AMD ordering, nnz(L)=%.3g' % info.lnz) if linsys == 0 else
                        'reduced-KKT Jacobi-PCG (the direct factorisation did not fit the time limit), %.1f PCG iterations per ADMM iteration' % (info.pcg_iters / max(info.iter, 1)), t_setup),
           'setup_s': t_setup}
    if done:
        out['iters_to_converge'] = int(info.iter)
        out['time_to_solution_ms'] = 1e3 * info.solve_time
        out['rho_updates'] = int(info.rho_updates)
    return out


def _cpu_sample_child(conn, P, q, A, l, u, settings, iters, linsys):
    try:
        import oracle
        from oracle import Oracle, SOLVED
        oracle.use_native()
        t0 = time.time()
        o = Oracle().setup(P, q, A, l, u, eps_abs=settings['eps_abs'], eps_rel=settings['eps_rel'], max_iter=iters, adaptive_rho_interval=settings['adaptive_rho_interval'],
                           check_termination=settings['check_termination'], linsys=linsys, **({'pcg_max_iter': 200, 'pcg_tol': 1e-7} if linsys else {}))
        ts = time.time() - t0
        _, _, info = o.solve()
        conn.send({'value': info.iter / info.solve_time, 'unit': 'ADMM iter/s', 'cores': 1, 'kind': 'port', 'setup_s': ts, 'converged': bool(info.status_val == SOLVED),
                   'sample': '%d cold-started ADMM iterations in %.1f s (%s), %s' % (info.iter, info.solve_time, 'run to convergence' if info.status_val == SOLVED else 'bounded sample',
                                                                                     'direct LDL\' KKT solve' if linsys == 0 else 'reduced-KKT Jacobi-PCG, %.1f PCG iterations per ADMM iteration' % (info.pcg_iters / max(info.iter, 1)))})
    except Exception as e:          # noqa: BLE001
        conn.send({'error': repr(e)})


def cpu_sample(P, q, A, l, u, settings, iters, linsys, limit_s):
    """A bounded sample of the oracle (iters ADMM iterations at most) in a child process with a hard time limit -- the CPU figure of an extra leg."""
    import multiprocessing as mp
    ctx = mp.get_context('fork')
    parent, child = ctx.Pipe(duplex=False)
    pr = ctx.Process(target=_cpu_sample_child, args=(child, P, q, A, l, u, settings, iters, linsys))
    pr.start()
    out = parent.recv() if parent.poll(limit_s) else None
    if out is None:
        pr.terminate()
    pr.join(5.0)
    return out if out is not None else {'value': None, 'sample': 'the oracle did not finish %d iterations within %.0f s' % (iters, limit_s)}


def config_leg(which, args, settings, osqp_amd, problems, torch):
    """BASELINE configs[2] (lasso) / configs[3] (portfolio) as a short leg of the DEFAULT line (the driver's record then carries them): first cold solve of
    a fresh handle, a few steady cold steps (rho reset), the roofline of what those solves launch, a bounded CPU-oracle sample."""
    import numpy as np
    if which == 'lasso':
        P, q, A, l, u = problems.lasso_qp(5000, 10000)
        name = 'BASELINE configs[2]: lasso-as-QP, 5k features x 10k samples, dense data block (problems.lasso_qp, seed 1)'
    else:
        P, q, A, l, u = problems.portfolio_qp(10000, 100)
        name = 'BASELINE configs[3]: portfolio factor model, 10k assets, 100 factors (problems.portfolio_qp, seed 1)'
    st = dict(settings); st['max_iter'] = 50000
    n, mm = len(q), len(l)
    h = osqp_amd.OSQP(algebra='hip')
    t0 = time.perf_counter(); h.setup(P, q, A, l, u, **st); t_setup = time.perf_counter() - t0
    t0 = time.perf_counter(); r = h.solve(); torch.cuda.synchronize(); first_ms = 1e3 * (time.perf_counter() - t0)
    s1 = h._solver.hip_stats()
    steps = []
    for _ in range(3):
        h.update_settings(rho=0.1)
        t0 = time.perf_counter(); r = h.solve(); torch.cuda.synchronize(); steps.append(1e3 * (time.perf_counter() - t0))
    stats = h._solver.hip_stats()
    probes, kb, pcg_bytes, pcg_ms, dom, dom_kernel, _, _, _, _, _ = measure_roofline(h._solver, stats, n, mm, args)
    out = {'workload': name + ': n=%d m=%d nnz(A)=%d nnz(P)=%d, eps %g' % (n, mm, A.nnz, P.nnz, settings['eps_abs']),
           'setup_s': t_setup, 'first_cold_solve_ms': first_ms, 'first_solve_woodbury_factorisations': int(s1.get('woodbury_factorisations', 0)),
           'first_solve_woodbury_factor_ms': s1.get('woodbury_factor_ms', 0.0), 'ms_per_step': median(steps), 'ms_per_step_each': [round(v, 3) for v in steps],
           'steady_step_woodbury_cache_hits': int(stats.get('woodbury_cache_hits', 0)), 'steady_step_woodbury_factorisations': int(stats.get('woodbury_factorisations', 0)),
           'status': r.info.status, 'admm_iters': int(r.info.iter), 'obj_val': r.info.obj_val, 'preconditioner': h._solver.hip_preconditioner(),
           'pcg_iters_per_admm_iter': stats['pcg_iters_total'] / max(r.info.iter, 1), 'kernel_launches_per_solve': stats['kernel_launches'],
           'roofline': {'bound': 'hbm', 'kernel': dom, 'ms_per_launch_group': probes[dom]['ms'], 'launches': probes[dom].get('launches', 1), 'bytes': kb[dom],
                        'achieved': probes[dom]['GBps'], 'unit': 'GB/s', 'peak': HBM_PEAK_GBS, 'frac': probes[dom]['GBps'] / HBM_PEAK_GBS,
                        'traffic': pmc_traffic(dom_kernel, 'lasso_5k_10k' if which == 'lasso' else 'portfolio_10k_100')}}      # (PMC bytes per launch group: profiles/*_pmc_<workload>_*)
    del h
    # CPU oracle: portfolio converges in seconds on the direct path; the lasso's dense block does not factorise in the budget: two PCG-path iterations
    out['cpu_baseline'] = cpu_sample(P, q, A, l, u, st, 20000, 0, 40.0) if which == 'portfolio' else cpu_sample(P, q, A, l, u, st, 2, 1, 40.0)
    if out['cpu_baseline'].get('value'):
        out['gpu_over_cpu_iter_rate'] = (r.info.iter / (median(steps) * 1e-3)) / out['cpu_baseline']['value']
    return out


def perturbed_resolves(m, q, l, u, steps, rho0, torch):
    """What the reference's parametric use is (update(q, l, u) + solve(), /root/reference/src/osqp/nn/torch.py:136-140): before every cold solve q, l, u are
    REDRAWN (q + 0.1 N(0,1); every row's interval shifted by 0.05 N(0,1): equality rows stay equalities, l <= u holds) and handed over with
    update_data_vec.  The handle keeps its launch history and graphs; the course of the solve changes with the data."""
    import numpy as np
    rng = np.random.default_rng(2024)
    upd, sol, its = [], [], []
    for _ in range(steps):
        qn = q + 0.1 * rng.standard_normal(len(q)); sh = 0.05 * rng.standard_normal(len(l))
        t0 = time.perf_counter(); m.update(q=qn, l=l + sh, u=u + sh); m.update_settings(rho=rho0); torch.cuda.synchronize(); t1 = time.perf_counter()
        r = m.solve(); torch.cuda.synchronize(); t2 = time.perf_counter()
        upd.append(1e3 * (t1 - t0)); sol.append(1e3 * (t2 - t1)); its.append(int(r.info.iter))
        assert r.info.status == 'solved', r.info.status
    m.update(q=q, l=l, u=u); m.update_settings(rho=rho0)
    return {'steps': steps, 'solve_ms_median': median(sol), 'solve_ms_mean': sum(sol) / len(sol), 'solve_ms_each': [round(v, 2) for v in sol], 'update_ms_median': median(upd),
            'admm_iters_each': its, 'what': 'q, l, u redrawn before every cold solve (update_data_vec, rho reset); launch history and graphs of the handle kept'}


def upstream_osqp_line(P, q, A, l, u, settings):
    """when the real `osqp` package (the reference's compiled C core) is importable on the timing host, time it on the same QP and label
    it "upstream osqp x.y.z"; it is not part of this repo and not expected on the GPU box (no network): then the line says so."""
    try:
        import importlib
        osqp = importlib.import_module('osqp')
        if 'osqp_amd' in (getattr(osqp, '__file__', '') or ''):
            raise ImportError('only this repo\'s own package answers to the name')
    except Exception as e:                       # noqa: BLE001
        return {'available': False, 'note': 'import osqp failed on this host (%s): the reference\'s C core is not installed; cpu_baseline is the oracle (kind = "port")' % type(e).__name__}
    try:
        import scipy.sparse as sp
        m = osqp.OSQP()
        m.setup(sp.triu(P).tocsc(), q, A, l, u, eps_abs=settings['eps_abs'], eps_rel=settings['eps_rel'], max_iter=settings['max_iter'], verbose=False)
        t0 = time.perf_counter(); r = m.solve(); dt = time.perf_counter() - t0
        return {'available': True, 'label': 'upstream osqp %s' % getattr(osqp, '__version__', '?'), 'status': r.info.status, 'iter': int(r.info.iter), 'solve_s': dt, 'it_per_s': r.info.iter / dt}
    except Exception as e:                       # noqa: BLE001
        return {'available': True, 'error': repr(e)}


def measure_roofline(s, stats, n, mm, args):
    """Live timing of the hot-path kernels on the solver's own data (hipEvent pairs on the solver's stream, osqp_hip_time_kernel) against their
    algorithmic bytes.  Returns (probes, kb, pcg_bytes, pcg_ms, dom, dom_kernel, survey_pcg_bytes, streamed, f1, fused, f1_D)."""
    nnzA, nnzB = int(stats['nnzA']), int(stats['nnzB'])
    fused = bool(stats.get('pcg_fused', 0))
    f1 = int(stats.get('pcg_fused', 0)) == 2          # one launch per PCG iteration
    f1_D = int(stats.get('f1_replicas', 0))
    sA, sB = spmv_bytes(nnzA, mm, n), spmv_bytes(nnzB, n, n + mm)
    # B_pcg = B_P + B_A + B_At + 104 n  (the reference algorithm's PCG iteration: three SpMVs + 13 n-vector passes)
    nnzP_full = nnzB - nnzA
    survey_pcg_bytes = spmv_bytes(nnzP_full, n, n) + spmv_bytes(nnzA, mm, n) + spmv_bytes(nnzA, n, mm) + 104 * n
    # algorithmic bytes per launch (DESIGN.md "Kernels"): SpMV formula + the fused epilogue / extra vectors
    if f1:
        # what the F1 kernel itself has to move: A once (8-byte values + one packed 32-bit index word per entry, row pointers, rho,
        # 16-bit column pointers of the windows ~ 2 bytes per column and replica), P + sigma I once (CSR), and per column: Minv, r, pu,
        # s, D replicas read; p, x~ read; s, r, p, x~, pu and D replicas written
        f1_bytes = 12 * nnzA + 4 * (mm + 1) + 8 * mm + 2 * f1_D * n + 12 * nnzP_full + 4 * (n + 1) + 8 * n * (4 + f1_D + 2 + 5 + f1_D)
        pcg_kernels = {'F1 one PCG iteration per launch (k_slot1 phase F)': (14, f1_bytes)}
        seq_id, dom, dom_kernel = 16, 'F1 one PCG iteration per launch (k_slot1 phase F)', 'k_f1_probe'
    elif fused:     # two kernels per PCG iteration
        pcg_kernels = {
            # SpMV(A) applied to Minv.*s (the gathered vector, counted in sA) with the epilogue t = t - alpha rho S (+ rho, t read:
            # 16m; t written = sA's output), + the vector update: u p r s Minv x~ read, p x~ r u' written (10 x 8n)
            'K1F spmv A + pcg vector update (k_k1f)': (11, sA + 2 * 8 * mm + 10 * 8 * n),
            # SpMV(B) whose output is s (w is never stored), + s and Minv read (16n), + Minv.*s written (8n)
            'K2F spmv B + s, Minv.*s, <s,u> (k_k2f)': (12, sB + 3 * 8 * n),
        }
        seq_id, dom, dom_kernel = 10, 'K2F spmv B + s, Minv.*s, <s,u> (k_k2f)', 'k_k2f'
    else:
        pcg_kernels = {
            'K1 spmv A (t=rho.*(A u))': (0, sA + 8 * mm),
            'K2 spmv B (w=B[u;t], <w,u>)': (1, sB),
            'Kv pcg vector update': (2, 12 * 8 * n),
        }
        seq_id, dom, dom_kernel = 6, 'K2 spmv B (w=B[u;t], <w,u>)', 'k_k2'
    if f1:
        # F1 form, k + 2 launches per ADMM iteration: no pass over B.  KA streams A once (12 bytes per entry + row pointers), reads
        # l, u, rho, rho_inv, z, y, z~_prev and writes y, dy, z, z~, v, A x_g, t0 per row (14 m-vector passes), reads x~ (window), x, x~_prev, q and
        # writes x, dx, x_g per column, streams P + sigma I once, and leaves the slices of r_0 and of rhs (2 D n-vector passes written);
        # the chunk's first launch runs its transposed passes only (v, t0, x, x_g, q read)
        nzP = nnzB - nnzA
        ka_bytes = 12 * nnzA + 4 * (mm + 1) + 8 * mm * 14 + 12 * nzP + 4 * (n + 1) + 8 * n * (4 + 3 + 2 * f1_D)
        other = {
            'KA z~ = A x~, z / y / x update, slices of r_0 and rhs (k_slot1 phase KA)': (17, ka_bytes),
            'chunk start: slices of r_0 and rhs from the vectors in memory (k_slot1 phase KB)': (18, 12 * nnzA + 8 * mm * 2 + 12 * nzP + 4 * (n + 1) + 8 * n * (3 + 2 * f1_D)),
        }
    else:
        other = {
            # (+ the extrapolated PCG start: KB resets x~ to it (8n written); KA reads the previous z~ and x~_prev and writes
            #  A xg, xg, x~_prev: 8m + 8n read, 8m + 16n written)
            'KB rhs + pcg start': (3, sB + 8 * mm + 8 * (5 * n)),
            'KA A x~ + z,y,x update + next PCG start': (4, sA + 8 * (11 * mm) + 8 * (6 * n)),
        }
    probes = {}
    for name, (which, nbytes) in {**pcg_kernels, **other}.items():
        probes[name] = {'ms_same_kernel_repeat': s.hip_time_kernel(which, args.probe_reps), 'bytes': nbytes}
    # in-sequence times: T(one PCG iteration as a solve runs it) minus T(the sequence without the kernel); this is what
    # a solve pays (the kernels evict each other's matrix from L2)
    pcg_ms = s.hip_time_kernel(seq_id, args.probe_reps)
    if f1:
        # every probe = two consecutive iterations (the double-buffered vectors alternate as in a solve).  16: F launches of the SLOT KERNEL
        # ITSELF (phase record, scalars from the fold of the previous launch's partials, stopping test that never fires): what a launch costs
        # inside a solve -- the figure the roofline uses.  15: the same iteration in a kernel that holds nothing but the F phase, fixed
        # scalars; 14: that kernel without the fold.
        pcg_ms *= 0.5
        probes[dom]['ms'] = pcg_ms
        probes[dom]['ms_f_only_kernel'] = 0.5 * s.hip_time_kernel(15, args.probe_reps)
        probes[dom]['ms_f_only_kernel_without_scalar_fold'] = 0.5 * probes[dom].pop('ms_same_kernel_repeat')
    elif fused:   # the "sequence without the kernel" is the other kernel alone (L2-hot, so this is an upper bound)
        names = list(pcg_kernels)
        for name, other_name in zip(names, names[::-1]):
            probes[name]['ms'] = max(pcg_ms - probes[other_name]['ms_same_kernel_repeat'], 1e-6)
    else:
        for name, which in zip(list(pcg_kernels), (8, 7, 9)):
            probes[name]['ms'] = max(pcg_ms - s.hip_time_kernel(which, args.probe_reps), 1e-6)
    for name in other:
        probes[name]['ms'] = probes[name]['ms_same_kernel_repeat']
    for name in probes:
        probes[name]['GBps'] = probes[name]['bytes'] / (probes[name]['ms'] * 1e-3) / 1e9
    wdirect = int(stats.get('woodbury_direct', 0))
    if wdirect:
        # The Woodbury direct mode runs NO PCG iteration: what a solve launches per ADMM iteration is (2) the two-launch form k_wbx_x + k_wbx_y
        # (portfolio: a dense r x 64 tile of A_L resp. S^-1 A_L per workgroup from LDS), or (1) KB, the three kernels of M^-1 = K^-1
        # (k_wb_p1 over the long rows, k_wb_gemv with S^-1, k_wb_p3 over their transpose) and KA.  The roofline names THAT.
        r = int(stats.get('woodbury_rows', 0))
        if wdirect == 2:
            G = (n + 63) // 64
            wb_bytes = 2 * G * 128 * 64 * 8 + 8 * n * 12 + 8 * mm * 8          # both tiles as stored (zero-padded to 128 rows) + the n- and m-vector passes of X and Y
            ms_it = s.hip_time_kernel(20, max(50, args.probe_reps))
            one = int(stats.get('woodbury_one_launch', 0))
            if one:      # one launch per ADMM iteration: the tile of A_L once (no S^-1 A_L tile), S^-1 (128 x 128 padded) per workgroup, the vector passes
                wb_bytes = G * 128 * 64 * 8 + G * 128 * 128 * 8 + 8 * n * 12 + 8 * mm * 8
            name = 'Woodbury direct mode, one ADMM iteration = ONE launch (k_wbz: fold of g, h = S^-1 g from registers, long rows, own columns, next right-hand side)' if one else 'Woodbury direct mode, one ADMM iteration = k_wbx_y + k_wbx_x (two launches)'
            probes = {name: {'ms': ms_it, 'ms_same_kernel_repeat': ms_it, 'bytes': wb_bytes, 'GBps': wb_bytes / (ms_it * 1e-3) / 1e9, 'launches': 1 if one else 2}}
            dom, dom_kernel = name, ('k_wbz' if one else 'k_wbx_x')
        else:
            nzL = nnzA                                                         # (the long rows carry nearly all of A in this form)
            cd = int(stats.get('woodbury_dual_cols', 0))                       # column-space form: the dense system is cd x cd (OSQPHipPolicy::woodbury_dual)
            order = cd or r
            wb_bytes = 8 * nzL + 8 * order * order + 8 * nzL + 8 * (4 * n + 2 * r)     # A_L once (values only: consecutive columns), the inverse, A_L' once, the vectors
            if int(stats.get('woodbury_fused_iteration', 0)):
                # the fused iteration: A_L streamed twice (transposed pass with the combined vector, row pass that also updates z / y of the dense rows), T^-1 once,
                # the m- and n-vector passes of the seven launches -- a whole ADMM iteration, no KB / KA launch beside it
                wb_bytes = 2 * 8 * nzL + 8 * order * order + 8 * (12 * n + 16 * mm)
                ms_it = s.hip_time_kernel(23, max(20, args.probe_reps // 4))
                fmode = int(stats.get('woodbury_fused_iteration', 0))
                dense = fmode >= 2
                name = ('Woodbury direct mode in column space, FUSED ADMM iteration, dense block held dense: k_wbf_rb + k_wbf_gd + k_wbf_gr + k_wbd_gemv (T^-1, %d x %d) + k_wbf_td + k_wbf_x + k_wbf_s2 (seven launches, no KB / KA)' % (cd, cd)) if fmode == 3 else \
                       ('Woodbury direct mode in column space, FUSED ADMM iteration, dense block held dense: k_wbf_r + k_wbf_beta + k_wbf_gd + k_wbf_gr + k_wbd_gemv (T^-1, %d x %d) + k_wbf_td + k_wbf_x + k_wbf_s (eight launches, no KB / KA)' % (cd, cd)) if dense else \
                       ('Woodbury direct mode in column space, FUSED ADMM iteration: k_wbf_r + k_wbf_beta + k_wbf_g + k_wbd_gemv (T^-1, %d x %d) + k_wbf_t + k_wbf_x + k_wbf_s (seven launches, no KB / KA)' % (cd, cd))
                probes = {name: {'ms': ms_it, 'ms_same_kernel_repeat': ms_it, 'bytes': wb_bytes, 'GBps': wb_bytes / (ms_it * 1e-3) / 1e9, 'launches': 8 if fmode == 2 else 7}}
                kb = {name: wb_bytes}
                group = 'k_wbf_rb+k_wbf_gd+k_wbf_gr+k_wbd_gemv+k_wbf_td+k_wbf_x+k_wbf_s2' if fmode == 3 else 'k_wbf_g'      # (roofline.traffic: PMC bytes summed over the launch group)
                return probes, kb, wb_bytes, ms_it, name, group, survey_pcg_bytes, None, False, fused, 0
            ms_ch = s.hip_time_kernel(21, max(20, args.probe_reps // 4))
            name = ('Woodbury direct mode in column space, M^-1 = K^-1: k_wbd_beta + k_wbd_g + k_wbd_gemv (T^-1, %d x %d) + k_wbd_t + k_wbd_fin (five launches per ADMM iteration)' % (cd, cd)) if cd else \
                   'Woodbury direct mode, M^-1 = K^-1: k_wb_p1 + k_wb_gemv + k_wb_p3 (three launches per ADMM iteration)'
            probes[name] = {'ms': ms_ch, 'ms_same_kernel_repeat': ms_ch, 'bytes': wb_bytes, 'GBps': wb_bytes / (ms_ch * 1e-3) / 1e9, 'launches': 5 if cd else 3}
            dom, dom_kernel = name, ('k_wbd_gemv' if cd else 'k_wb_gemv')
        kb = {nm: probes[nm]['bytes'] for nm in probes}
        return probes, kb, kb[dom], probes[dom]['ms'], dom, dom_kernel, survey_pcg_bytes, None, False, fused, 0
    kb = {name: probes[name]['bytes'] for name in probes}
    pcg_bytes = sum(kb[k] for k in pcg_kernels)
    if f1:
        # one launch = one PCG iteration of the reference algorithm: the contract's algorithmic bytes per launch are SURVEY §8(d)'s
        # B_pcg; the kernel's own (smaller) traffic model is reported next to it as streamed_bytes / frac_streamed
        probes[dom]['streamed_bytes'] = kb[dom]; probes[dom]['GBps_streamed'] = probes[dom]['GBps']
        kb[dom] = survey_pcg_bytes; probes[dom]['bytes'] = survey_pcg_bytes
        probes[dom]['GBps'] = survey_pcg_bytes / (probes[dom]['ms'] * 1e-3) / 1e9
        streamed = pcg_bytes; pcg_bytes = survey_pcg_bytes
    return probes, kb, pcg_bytes, pcg_ms, dom, dom_kernel, survey_pcg_bytes, (streamed if f1 else None), f1, fused, f1_D


def self_launch(ngpus, script=None):
    """`python bench.py --gpus N` with N > 1 and no launcher around it (WORLD_SIZE unset): start the N ranks ourselves -- the same
    torch.distributed.run command line the task statement gives, one process per GPU, rendezvous on 127.0.0.1 -- and hand its exit code on."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # dmabuf IPC only on this driver (RCCL / tensor sharing across processes)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(ngpus), '--master-addr', '127.0.0.1',
           '--master-port', str(port), script or os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--n', '--vars', dest='n', type=int, default=100000, help='variables (m = 2n, nnz(A) = 10n, nnz(P) = 2n); --vars: the spelling torch.distributed.run passes through')
    ap.add_argument('--carry-rho', action='store_true', help='keep the rho a solve ended with for the next step (the solver object\'s natural behaviour) instead of restarting every step from the setting')
    ap.add_argument('--config', default='banded', choices=['banded', 'shuffled', 'unstructured', 'mixed', 'lasso', 'portfolio'],
                    help="banded = BASELINE configs[1] (the headline); shuffled = the same QP with its variables and constraints randomly renumbered (the band is "
                         "there but hidden: the engine has to find it, OSQPHipPolicy::reorder); unstructured = the same sizes with columns drawn from the whole row (GB/s only, "
                         "SURVEY 8d); lasso = configs[2] (5k features x 10k samples, dense data block); portfolio = configs[3] (10k assets, 100 factors)")
    ap.add_argument('--eps', type=float, default=1e-6)
    ap.add_argument('--cpu-seconds', type=float, default=40.0, help='CPU-baseline budget (0 disables)')
    ap.add_argument('--probe-reps', type=int, default=200)
    ap.add_argument('--dist-backend', default='nccl', help="'nccl' (= RCCL, the production path) or 'gloo' (code-path test)")
    ap.add_argument('--single-device', action='store_true', help='test mode: every rank uses GPU 0 (one-GPU boxes)')
    ap.add_argument('--batch', type=int, default=4096, help='BASELINE configs[4]: MPC QPs solved through the sharded batch path and reported as config.batch (0 disables)')
    ap.add_argument('--batch-steps', type=int, default=5)
    ap.add_argument('--batch-cpu', type=int, default=1, help='time the all-cores CPU baseline of the batch too (N = 1 only)')
    ap.add_argument('--hbm-n', type=int, default=1000000, help='N = 1, headline config only: also time the dominant kernel on the same generator at this many variables '
                                                               '(a working set beyond the 256 MiB Infinity Cache) and report it as roofline.hbm_resident (0 disables)')
    ap.add_argument('--jacobi-leg', type=int, default=1, help='lasso / portfolio: also solve once with the plain Jacobi preconditioner (config.jacobi_only); 0 for profiling runs, whose kernel statistics it would dominate')
    ap.add_argument('--extra-legs', type=int, default=1, help='N = 1, headline config only: short legs of BASELINE configs[2] (lasso) and configs[3] (portfolio) reported as config.lasso / config.portfolio, and the perturbed re-solve of the headline QP (0 disables)')
    ap.add_argument('--unstructured-leg', type=int, default=1, help='N = 1, headline config only: also solve the unstructured variant of the same sizes and report its PCG iteration as roofline.unstructured (0 disables)')
    args = ap.parse_args()
    warnings.simplefilter('ignore')
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(self_launch(args.gpus))

    import numpy as np
    import torch
    import torch.distributed as dist
    import osqp_amd
    import problems

    rank = int(os.environ.get('RANK', 0)); world = int(os.environ.get('WORLD_SIZE', 1)); local = int(os.environ.get('LOCAL_RANK', 0))
    assert world == args.gpus, 'WORLD_SIZE = %d but --gpus %d' % (world, args.gpus)
    if args.single_device:
        local = 0
    # OSQP_BENCH_HOSTSIM=1 (tests/test_bench_launch.py ONLY): the launch / sharding / gather logic of this script on a machine without a GPU --
    # the engine's host driver linked to the plain-loop device simulator of tests/hostsim, CPU tensors, gloo.  Nothing it prints is a measurement
    # (the line says so in `data`); the product path has no such switch: without it a missing HIP device fails in osqp_setup.
    hostsim = bool(os.environ.get('OSQP_BENCH_HOSTSIM'))
    if hostsim:
        sys.path.insert(0, os.path.join(ROOT, 'tests'))
        from hostsim_util import hostsim as _hostsim_ctx
        _ctx = _hostsim_ctx(); _ctx.__enter__()
        args.dist_backend = 'gloo'; args.cpu_seconds = 0.0; args.hbm_n = 0
    # OSQP_BENCH_FORCE_DIST=1: initialise the process group (RCCL) also for one rank -- exercises init / barrier / all_gather on a 1-GPU box
    use_dist = world > 1 or bool(os.environ.get('OSQP_BENCH_FORCE_DIST'))
    if not hostsim:
        torch.cuda.set_device(local)
    if use_dist:
        if args.dist_backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local))     # backend "nccl" is RCCL on ROCm
        else:
            dist.init_process_group(args.dist_backend)

    n = args.n
    # replicas: every rank solves the same QP (identical work per GPU)
    if args.config == 'banded':
        P, q, A, l, u = problems.banded_qp(n, seed=12345); wl_tag = 'banded_n%d' % n
        wl_name = 'BASELINE configs[1]: single QP n=%d m=%d nnz(A)=%d nnz(P)=%d (problems.banded_qp, seed 12345)'
    elif args.config == 'shuffled':
        P, q, A, l, u = problems.banded_qp(n, seed=12345); wl_tag = 'shuffled_n%d' % n
        _rng = np.random.default_rng(99); _pc, _pr = _rng.permutation(n), _rng.permutation(A.shape[0])
        P = P[_pc][:, _pc].tocsc(); A = A[_pr][:, _pc].tocsc(); P.sort_indices(); A.sort_indices(); q, l, u = q[_pc], l[_pr], u[_pr]
        wl_name = 'configs[1] with RANDOMLY RENUMBERED variables and constraints (hidden band): single QP n=%d m=%d nnz(A)=%d nnz(P)=%d (problems.banded_qp + permutation, seeds 12345 / 99)'
    elif args.config == 'unstructured':
        P, q, A, l, u = problems.banded_qp(n, window=n, seed=12345); wl_tag = 'unstructured_n%d' % n
        wl_name = 'configs[1] sizes with UNSTRUCTURED columns (SURVEY 8d: GB/s only): single QP n=%d m=%d nnz(A)=%d nnz(P)=%d (problems.banded_qp, window = n)'
    elif args.config == 'mixed':
        P, q, A, l, u = problems.banded_qp(n, seed=12345, long_range=0.02); wl_tag = 'mixed_n%d' % n
        wl_name = 'configs[1] with 2 %% of the entries of A moved to columns drawn from the whole range (band + long-range couplings): single QP n=%d m=%d nnz(A)=%d nnz(P)=%d (problems.banded_qp, long_range = 0.02)'
    elif args.config == 'lasso':
        P, q, A, l, u = problems.lasso_qp(5000, 10000); wl_tag = 'lasso_5k_10k'
        wl_name = 'BASELINE configs[2]: lasso-as-QP, 5k features x 10k samples, dense data block: n=%d m=%d nnz(A)=%d nnz(P)=%d (problems.lasso_qp, seed 1)'
    else:
        P, q, A, l, u = problems.portfolio_qp(10000, 100); wl_tag = 'portfolio_10k_100'
        wl_name = 'BASELINE configs[3]: portfolio factor model, 10k assets, 100 factors: n=%d m=%d nnz(A)=%d nnz(P)=%d (problems.portfolio_qp, seed 1)'
    n = len(q)
    settings = dict(eps_abs=args.eps, eps_rel=args.eps, max_iter=50000 if args.config in ('lasso', 'portfolio') else 20000, check_termination=25,
                    adaptive_rho_interval=50, scaling=10, warm_starting=False, verbose=False, device=local)
    m = osqp_amd.OSQP(algebra='hip')
    t0 = time.time()
    m.setup(P, q, A, l, u, **settings)
    t_setup = time.time() - t0

    def barrier():
        if use_dist:
            dist.barrier()
        if not hostsim:
            torch.cuda.synchronize()
    sync = (lambda: None) if hostsim else torch.cuda.synchronize

    # Every step is a TRUE cold solve: x, z, y restart from zero (warm_starting = False) and rho is put back to the setting's value
    # (the solver object, like the reference's -- adapt_rho writes settings.rho, _osqp.py:923-930 -- would otherwise carry the rho a
    # solve ended with into the next, and the iteration count of a step would depend on how many steps came before it; --carry-rho
    # restores that behaviour).  The handle's first solve also pays the one-time capture of the launch graphs: timed on its own.
    rho0 = 0.1
    def cold_solve():
        if not args.carry_rho:
            m.update_settings(rho=rho0)
        return m.solve()
    first_ms, first_iters = None, None
    step_iters, step_ms, step_event_ms = [], [], []
    for w in range(args.warmup):
        tw = time.perf_counter(); rw = cold_solve(); sync()
        if w == 0:
            first_ms, first_iters = 1e3 * (time.perf_counter() - tw), int(rw.info.iter)
    barrier()
    t0 = time.perf_counter()
    iters = 0
    res = None
    for _ in range(args.steps):
        ts = time.perf_counter()
        res = cold_solve()
        iters += res.info.iter
        step_iters.append(int(res.info.iter)); step_ms.append(1e3 * (time.perf_counter() - ts))
        step_event_ms.append(m._solver.hip_stats()['gpu_solve_ms'])      # hipEvent pair on the solver's stream around osqp_solve
    barrier()
    elapsed = time.perf_counter() - t0
    stats = m._solver.hip_stats()

    # whole-job aggregate: total ADMM iterations / max-over-ranks time; final status/objective gather over RCCL
    rec = torch.tensor([float(res.info.status_val), float(res.info.iter), res.info.obj_val, res.info.prim_res, res.info.dual_res,
                        elapsed, float(iters)], dtype=torch.float64, device='cuda' if (args.dist_backend == 'nccl' and not hostsim) else 'cpu')
    if use_dist:
        allrec = [torch.empty_like(rec) for _ in range(world)]
        dist.all_gather(allrec, rec)
        allrec = torch.stack(allrec).cpu().numpy()
    else:
        allrec = rec.cpu().numpy()[None, :]
    tmax = float(allrec[:, 5].max()); total_iters = float(allrec[:, 6].sum())

    # BASELINE configs[4] through the sharded batch path -- the workload north_star really shards (one contiguous block of problems per GPU,
    # one all_gather of the records) -- on every rank, at any N: `value` stays the replica metric, the batch travels as config.batch so that
    # a scaling run of this script shows its strong-scaling curve (reference analogue: /root/reference/src/osqp/nn/torch.py:200-224)
    batch_out = None
    if args.batch > 0 and args.config == 'banded' and (args.dist_backend == 'nccl' or hostsim):
        import bench_batch
        if hostsim:
            batch_out = bench_batch.measure_sharded_host(args.batch, args.batch_steps, rank, world, use_dist)
        else:
            batch_out = bench_batch.measure_sharded_device(args.batch, args.batch_steps, 3, rank, world, local, use_dist)

    if rank == 0:
        mm = len(l)
        s = m._solver
        dev_state = None if hostsim else device_state(local)
        if hostsim:
            probes = kb = dom = dom_kernel = streamed = None; pcg_bytes = pcg_ms = survey_pcg_bytes = 0; f1 = fused = False; f1_D = 0
        else:
            probes, kb, pcg_bytes, pcg_ms, dom, dom_kernel, survey_pcg_bytes, streamed, f1, fused, f1_D = measure_roofline(s, stats, n, mm, args)
        tts_ms = 1e3 * tmax / args.steps
        out = {
            'metric': 'ADMM iterations/sec, n=%d m=%d nnz(A)=%d %s (indirect PCG)' % (n, mm, A.nnz, 'sparse QP' if args.config in ('banded', 'shuffled', 'unstructured', 'mixed') else args.config + ' QP'),
            'value': total_iters / tmax, 'unit': 'ADMM iter/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': 1e3 * tmax / args.steps, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f64', 'data': 'synthetic' if not hostsim else 'synthetic -- HOST SIMULATOR (OSQP_BENCH_HOSTSIM test mode): launch logic only, NOT a measurement',
            'config': {'workload': (wl_name % (n, mm, A.nnz, P.nnz)) + ', eps_abs=eps_rel=%g, indirect PCG, one replica per GPU' % args.eps,
                       'time_to_solution_ms': tts_ms, 'admm_iters_per_solve': int(res.info.iter), 'status': res.info.status, 'obj_val': res.info.obj_val,
                       # what a step is, exactly: x, z, y restart from zero and rho restarts from the setting (see cold_solve above); the
                       # handle's first solve, which also captures the launch graphs, is reported separately
                       'rho_carried_between_steps': bool(args.carry_rho), 'first_cold_solve_ms': first_ms, 'first_cold_solve_admm_iters': first_iters,
                       'mean_admm_iters_per_step': sum(step_iters) / max(len(step_iters), 1), 'admm_iters_per_step': step_iters,
                       'ms_per_step_each': [round(v, 3) for v in step_ms],
                       # SURVEY 8(d)'s timing form next to the contract's wall-clock mean: a hipEvent pair on the solver's stream around every
                       # osqp_solve (OSQPHipStats::gpu_solve_ms), median over the timed steps; and the wall-clock median / spread of the same steps
                       'solve_ms_hipevent_median': median(step_event_ms), 'solve_ms_hipevent_each': [round(v, 3) for v in step_event_ms],
                       'ms_per_step_median': median(step_ms), 'ms_per_step_max_minus_min': (max(step_ms) - min(step_ms)) if step_ms else None,
                       'prim_res': res.info.prim_res, 'dual_res': res.info.dual_res, 'rho_updates': int(res.info.rho_updates),
                       'pcg_iters_per_admm_iter': stats['pcg_iters_total'] / max(res.info.iter, 1), 'pcg_budget_limited_iters': int(stats['pcg_unconverged']),
                       'cg_cap_escalations': int(stats.get('cg_cap_escalations', 0)), 'slot_topups': int(stats.get('slot_topups', 0)),
                       'windowed_row_blocks': '%d of %d' % (int(stats.get('windowed_blocks', 0)), int(stats.get('row_blocks', 0))),
                       'reordered': bool(stats.get('reordered', 0)), 'reorder_ms': stats.get('reorder_ms', 0.0),
                       'preconditioner': s.hip_preconditioner(), 'woodbury_factorisations_last_solve': int(stats.get('woodbury_factorisations', 0)),
                       'woodbury_factor_ms_last_solve': stats.get('woodbury_factor_ms', 0.0),
                       # (rho updates of the last solve served by an inverse the handle had computed for the same rho_bar in an EARLIER solve -- every step of this
                       #  bench restarts from the setting's rho and walks the same rho values; first_cold_solve_ms is the figure without any cached inverse)
                       'woodbury_cache_hits_last_solve': int(stats.get('woodbury_cache_hits', 0)),
                       'pcg_kernels_per_iteration': 1 if f1 else (2 if fused else 3), 'f1_replicas': int(stats.get('f1_replicas', 0)), 'f1_far_columns': int(stats.get('f1_far_columns', 0)), 'kernel_launches_per_solve': stats['kernel_launches'], 'graph_launches_per_solve': stats['graph_launches'],
                       'setup_s': t_setup, 'per_rank': [{'status': int(r[0]), 'iter': int(r[1]), 'obj': r[2]} for r in allrec]},
            'device': dev_state,
            'roofline': None if hostsim else {'bound': 'hbm', 'kernel': dom + (' -- in solves this body runs as the K2F phase of k_slot_b' if (fused and not f1) else ''), 'achieved': probes[dom]['GBps'], 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': probes[dom]['GBps'] / HBM_PEAK_GBS, 'traffic': pmc_traffic(dom_kernel, wl_tag),
                         'bytes_per_launch': kb[dom], 'ms_per_launch': probes[dom]['ms'],
                         'pcg_iteration': {'bytes': pcg_bytes, 'ms': pcg_ms, 'GBps': pcg_bytes / (pcg_ms * 1e-3) / 1e9,
                                           'frac': pcg_bytes / (pcg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                           'bytes_survey_8d': survey_pcg_bytes,
                                           'frac_survey_8d': survey_pcg_bytes / (pcg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
                         'kernels': probes},
        }
        if f1:
            out['roofline']['streamed_bytes'] = streamed
            out['roofline']['frac_streamed'] = streamed / (pcg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
            out['roofline']['replicas'] = f1_D
            if out['roofline']['traffic']:
                out['roofline']['frac_traffic'] = out['roofline']['traffic'] / (pcg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS      # by the bytes the PMC counters saw
        if int(stats.get('woodbury_rows', 0)) > 0 and args.jacobi_leg:
            # the figure with the reference's literal preconditioner next to it: the same QP, same settings, plain Jacobi (OSQPHipPolicy::woodbury = 0)
            os.environ['OSQP_HIP_WOODBURY'] = '0'
            try:
                mj = osqp_amd.OSQP(algebra='hip'); mj.setup(P, q, A, l, u, **settings)
                tj = time.perf_counter(); rj = mj.solve(); torch.cuda.synchronize(); tj = time.perf_counter() - tj
                sj = mj._solver.hip_stats()
                out['config']['jacobi_only'] = {'preconditioner': mj._solver.hip_preconditioner(), 'first_cold_solve_ms': 1e3 * tj, 'admm_iters': int(rj.info.iter), 'status': rj.info.status,
                                                'pcg_iters_per_admm_iter': sj['pcg_iters_total'] / max(rj.info.iter, 1),
                                                'note': 'to compare with config.first_cold_solve_ms (%.1f ms, %s)' % (first_ms or 0.0, s.hip_preconditioner())}
                del mj
            finally:
                os.environ.pop('OSQP_HIP_WOODBURY', None)
        if batch_out is not None:
            bdata = batch_out.pop('_data')
            out['config']['batch'] = batch_out
            if args.batch_cpu and args.cpu_seconds > 0 and world == 1:
                import bench_batch
                cb = bench_batch.cpu_batch_baseline(*bdata)
                batch_out['cpu_baseline'] = cb
                batch_out['gpu_over_cpu_all_cores'] = batch_out['QP_per_s'] / cb['value']
        if args.hbm_n > n and args.config == 'banded' and world == 1 and f1 and args.cpu_seconds > 0:      # (--cpu-seconds 0: the profiling runs -- no extra legs)
            # The headline QP's working set (~50 MB) lives in the Infinity Cache: the fraction above is a cache-resident figure.  The same kernel on the
            # same generator at hbm_n variables (n = 1M: ~0.5 GB of matrices and vectors per launch) is the HBM-resident point of the roofline.
            Pb, qb, Ab, lb, ub = problems.banded_qp(args.hbm_n, seed=12345)
            mb = osqp_amd.OSQP(algebra='hip'); mb.setup(Pb, qb, Ab, lb, ub, **settings)
            rb = mb.solve(); sb = mb._solver; stb = sb.hip_stats()
            if int(stb.get('pcg_fused', 0)) == 2:
                nb_, mb_ = len(qb), len(lb)
                nzA, nzB = int(stb['nnzA']), int(stb['nnzB']); nzP = nzB - nzA; Db = int(stb.get('f1_replicas', 0))
                bytes_8d = spmv_bytes(nzP, nb_, nb_) + spmv_bytes(nzA, mb_, nb_) + spmv_bytes(nzA, nb_, mb_) + 104 * nb_
                bytes_own = 12 * nzA + 4 * (mb_ + 1) + 8 * mb_ + 2 * Db * nb_ + 12 * nzP + 4 * (nb_ + 1) + 8 * nb_ * (4 + Db + 2 + 5 + Db)
                ms_l = 0.5 * sb.hip_time_kernel(16, max(20, args.probe_reps // 4))
                tr = pmc_traffic('k_f1_probe', 'banded_n%d' % nb_)
                out['roofline']['hbm_resident'] = {
                    'workload': 'the same generator at n=%d m=%d nnz(A)=%d (problems.banded_qp, seed 12345): %.0f MB per launch, beyond the 256 MiB Infinity Cache' % (nb_, mb_, Ab.nnz, bytes_own / 1e6),
                    'kernel': dom, 'ms_per_launch': ms_l, 'bytes_per_launch': bytes_8d, 'achieved': bytes_8d / (ms_l * 1e-3) / 1e9, 'unit': 'GB/s',
                    'frac': bytes_8d / (ms_l * 1e-3) / 1e9 / HBM_PEAK_GBS, 'streamed_bytes': bytes_own, 'frac_streamed': bytes_own / (ms_l * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    'traffic': tr, 'frac_traffic': (tr / (ms_l * 1e-3) / 1e9 / HBM_PEAK_GBS) if tr else None,
                    'solve': {'status': rb.info.status, 'admm_iters': int(rb.info.iter), 'first_cold_solve_ms': 1e3 * rb.info.solve_time}}
            del mb
        if args.unstructured_leg and args.config == 'banded' and world == 1 and not hostsim and args.cpu_seconds > 0:
            # "also run an unstructured variant for GB/s only" -- the same sizes with every row's columns drawn from the
            # whole range: no permutation gives its row blocks a compact window, so the PCG iteration runs as the two-kernel
            # pair with global gathers.  One solve + the pair's launch time, in the driver's record next to the banded figure.
            Pu, qu, Au, lu, uu_ = problems.banded_qp(n, window=n, seed=12345)
            mu = osqp_amd.OSQP(algebra='hip'); mu.setup(Pu, qu, Au, lu, uu_, **settings)
            tu = time.perf_counter(); ru = mu.solve(); torch.cuda.synchronize(); tu = time.perf_counter() - tu
            tu2 = time.perf_counter(); mu.update_settings(rho=rho0); ru = mu.solve(); torch.cuda.synchronize(); tu2 = time.perf_counter() - tu2
            su = mu._solver; stu = su.hip_stats()
            form = int(stu.get('pcg_fused', 0))
            nzA, nzB = int(stu['nnzA']), int(stu['nnzB']); nzP = nzB - nzA
            b8d = spmv_bytes(nzP, n, n) + spmv_bytes(nzA, mm, n) + spmv_bytes(nzA, n, mm) + 104 * n
            ms_it = (0.5 * su.hip_time_kernel(16, args.probe_reps)) if form == 2 else su.hip_time_kernel(10 if form == 1 else 6, args.probe_reps)
            tru = None
            if form == 1:
                t1, t2 = pmc_traffic('k_k2f', 'unstructured_n%d' % n), pmc_traffic('k_k1f', 'unstructured_n%d' % n)
                tru = (t1 + t2) if (t1 and t2) else None
            out['roofline']['unstructured'] = {
                'workload': 'configs[1] sizes, columns drawn from the whole row (problems.banded_qp, window = n, seed 12345): n=%d m=%d nnz(A)=%d' % (n, mm, Au.nnz),
                'launches_per_pcg_iteration': {2: 1, 1: 2}.get(form, 3), 'ms_per_pcg_iteration': ms_it, 'bytes_per_pcg_iteration': b8d,
                'achieved': b8d / (ms_it * 1e-3) / 1e9, 'unit': 'GB/s', 'frac': b8d / (ms_it * 1e-3) / 1e9 / HBM_PEAK_GBS,
                'traffic': tru, 'windowed_row_blocks': '%d of %d' % (int(stu.get('windowed_blocks', 0)), int(stu.get('row_blocks', 0))),
                'solve': {'status': ru.info.status, 'admm_iters': int(ru.info.iter), 'first_cold_solve_ms': 1e3 * tu, 'second_cold_solve_ms': 1e3 * tu2,
                          'pcg_iters_per_admm_iter': stu['pcg_iters_total'] / max(ru.info.iter, 1)}}
            # the same matrix through the one-launch form on the explicit reduced matrix (OSQPHipPolicy::kform, off by default): in the record so that the
            # choice is a measured one -- K needs nnz(K) random gathers per product where the A / B pair needs nnz(A) + nnz(B)
            os.environ['OSQP_HIP_KFORM'] = '1'
            try:
                mk = osqp_amd.OSQP(algebra='hip'); tks = time.perf_counter(); mk.setup(Pu, qu, Au, lu, uu_, **settings); tks = time.perf_counter() - tks
                mk.solve(); mk.update_settings(rho=rho0)
                tk = time.perf_counter(); rk = mk.solve(); torch.cuda.synchronize(); tk = time.perf_counter() - tk
                stk = mk._solver.hip_stats()
                if int(stk.get('pcg_fused', 0)) == 3:
                    ms_k = 0.5 * mk._solver.hip_time_kernel(16, args.probe_reps)
                    out['roofline']['unstructured']['kform'] = {
                        'what': 'one launch per PCG iteration on the explicit K = P + sigma I + A\' rho A (k_slotk): k + 3 launches per ADMM iteration instead of 2 k + 4',
                        'nnz_K': int(stk.get('kform_nnz', 0)), 'launches_per_pcg_iteration': 1, 'ms_per_pcg_iteration': ms_k, 'frac': b8d / (ms_k * 1e-3) / 1e9 / HBM_PEAK_GBS,
                        'second_cold_solve_ms': 1e3 * tk, 'admm_iters': int(rk.info.iter), 'status': rk.info.status, 'setup_s': tks, 'kernel_launches_per_solve': stk['kernel_launches'],
                        'verdict': 'slower than the two-kernel pair: off by default (profiles/r06a_kform_gather_bench.txt: 4.2 M random gathers cost 18-24 us of L2 -> L1 line fills alone)'}
                del mk
            finally:
                os.environ.pop('OSQP_HIP_KFORM', None)
            del mu
        if args.extra_legs and args.config == 'banded' and world == 1 and not hostsim and args.cpu_seconds > 0:
            out['config']['perturbed_resolve'] = perturbed_resolves(m, q, l, u, 20, rho0, torch)
            out['config']['perturbed_resolve_ms'] = out['config']['perturbed_resolve']['solve_ms_median']
            for leg in ('lasso', 'portfolio'):
                try:
                    out['config'][leg] = config_leg(leg, args, settings, osqp_amd, problems, torch)
                except Exception as e:          # noqa: BLE001 -- an extra leg must not lose the headline line
                    out['config'][leg] = {'error': repr(e)}
        if args.cpu_seconds > 0 and world == 1:          # (the CPU baseline is timed at N = 1 only: the other ranks would wait 40 s at the barrier)
            cb = cpu_baseline(P, q, A, l, u, settings, args.cpu_seconds)
            cb['upstream_osqp'] = upstream_osqp_line(P, q, A, l, u, settings)      # "if `import osqp` happens to succeed on the box ..."; never required
            out['cpu_baseline'] = cb
            if cb.get('value'):
                out['config']['gpu_over_cpu_iter_rate'] = (total_iters / tmax / world) / cb['value']
            if 'iters_to_converge' in cb:
                # the iteration-rate ratio overstates the end-to-end ratio when the two sides need different iteration counts:
                # report time-to-solution on both sides and the engine's rate in units of the reference path's iterations
                out['config']['oracle_iters_to_converge'] = cb['iters_to_converge']
                out['config']['reference_equivalent_iters_per_s'] = cb['iters_to_converge'] / (tts_ms * 1e-3)
                out['config']['gpu_over_cpu_time_to_solution'] = cb['time_to_solution_ms'] / tts_ms
        print(json.dumps(out))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
