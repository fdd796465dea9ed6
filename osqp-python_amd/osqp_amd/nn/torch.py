"""Forward pass of the reference's torch layer (/root/reference/src/osqp/nn/torch.py:22-230) on the MI355X engine.

The reference keeps one ``osqp.OSQP`` object per batch element alive ACROSS forward calls -- the first call sets them up, every
later call only ``update()``s and ``solve()``s them (nn/torch.py:113-140, 200-224) -- and fans the elements out over joblib
threads.  Here one persistent engine handle plays that role:

  * P_val and A_val shared by the whole batch (1-D tensors): the handle is set up ONCE (``setup_count`` counts it); later
    forwards re-upload the matrix values only if they changed (``osqp_update_data_mat``: device-side re-assembly) and solve
    the whole batch with ONE kernel launch (one workgroup per problem: the reference's update(q,l,u)+solve() per element).
    q / l / u that live on the GPU (torch ROCm tensors) are handed over ZERO-COPY by device pointer on torch's current
    stream (``osqp_hip_batch_solve_device``) and the solution is produced on the device;
  * with ``torch.distributed`` initialised (world size > 1) the batch is block-partitioned over the ranks, every rank solves its
    share on its own GPU, and the rows are all-gathered (``osqp_amd.sharded``) -- the multi-GPU form of the reference's thread pool;
  * per-element P_val / A_val (2-D tensors, :128-157, 184-217): the SAME single launch -- every workgroup reads its own element's matrix values, assembled
    and equilibrated per element by a launch in front of it (osqp_hip_batch_solve_mat); only a problem too large for one workgroup falls back to the
    single-QP handle, one element after the other.

Like the reference, a batch element that is not solved raises RuntimeError (:158-162).  Backward (adjoint derivatives,
:233-290) is out of scope of this engine: the returned tensor carries no grad_fn.
"""
import numpy as np
import scipy.sparse as spa
import torch
from torch.nn import Module

import osqp_amd
from osqp_amd import sharded


def _np(t):
    return t.detach().cpu().double().numpy()


def _distributed():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


class OSQP(Module):
    def __init__(self, P_idx, P_shape, A_idx, A_shape, eps_rel=1e-5, eps_abs=1e-5, verbose=False, max_iter=10000, algebra='hip', solver_type='indirect'):
        super().__init__()
        self.P_idx, self.P_shape, self.A_idx, self.A_shape = P_idx, P_shape, A_idx, A_shape
        self.eps_rel, self.eps_abs, self.verbose, self.max_iter = eps_rel, eps_abs, verbose, max_iter
        self.algebra, self.solver_type = algebra, solver_type
        self.n, self.m = P_shape[0], A_shape[0]
        self._solver = None          # the persistent handle (reference: the `solvers` list kept across forwards)
        self._Pv = self._Av = None   # matrix values currently on the handle
        self._device = None
        self._triu_pick = None       # positions of the upper-triangle entries of P inside P_val
        self.setup_count = 0         # number of osqp_setup calls made by this layer (1 after any number of same-structure forwards)
        self.last_dual = None

    # ------------------------------------------------------------------ the persistent handle
    def _matrices(self, P_val, A_val):
        P = spa.csc_matrix((P_val, self.P_idx), shape=self.P_shape)
        A = spa.csc_matrix((A_val, self.A_idx), shape=self.A_shape)
        return P, A

    def _handle(self, P_val, A_val, q0, l0, u0, device=0):
        """The set-up solver for these matrix values: created on first use, afterwards only update()d (nn/torch.py:136-140)."""
        if self._solver is None or self._device != device:
            P, A = self._matrices(P_val, A_val)
            if self._triu_pick is None:
                tag = spa.triu(self._matrices(np.arange(1, len(P_val) + 1, dtype=float), A_val)[0], format='csc')
                self._triu_pick = tag.data.astype(int) - 1
            s = osqp_amd.OSQP(algebra=self.algebra)
            s.setup(P, q0, A, l0, u0, solver_type=self.solver_type, verbose=self.verbose, eps_abs=self.eps_abs, eps_rel=self.eps_rel,
                    max_iter=self.max_iter, warm_starting=False, device=device)
            self._solver, self._device = s, device
            self._Pv, self._Av = np.array(P_val, dtype=float), np.array(A_val, dtype=float)
            self.setup_count += 1
        elif not (np.array_equal(self._Pv, P_val) and np.array_equal(self._Av, A_val)):
            self._solver.update(Px=np.asarray(P_val, dtype=float)[self._triu_pick], Ax=np.asarray(A_val, dtype=float))
            self._Pv, self._Av = np.array(P_val, dtype=float), np.array(A_val, dtype=float)
        return self._solver

    # ------------------------------------------------------------------ forward
    def forward(self, P_val, q_val, A_val, l_val, u_val):
        params = [P_val, q_val, A_val, l_val, u_val]
        for p in params:
            assert p.ndimension() <= 2, 'Unexpected number of dimensions'
        dtype, device = q_val.dtype, q_val.device
        batched = [p.ndimension() == 2 for p in params]
        nb = max([p.size(0) for p, b in zip(params, batched) if b], default=1)
        shared = not batched[0] and not batched[2]
        Pn, An = _np(P_val), _np(A_val)                                                # (small: the matrix VALUES, once per forward)
        rank, world = _distributed()
        if shared and q_val.is_cuda:                                                   # data on the GPU: zero-copy (one rank, or this rank's share)
            out = self._forward_device(Pn, An, q_val, l_val, u_val, nb, rank, world)
            if out is not None:
                return out if any(batched) else out.squeeze(0)
        bc = lambda a, k: a if a.ndim == 2 else np.broadcast_to(a, (nb, k))          # nn/torch.py:184-188
        qn, ln, un = bc(_np(q_val), self.n), bc(_np(l_val), self.m), bc(_np(u_val), self.m)
        if shared:                                                                      # shared matrices: batched kernel
            dev_index = (device.index or 0) if q_val.is_cuda else (torch.cuda.current_device() if torch.cuda.is_available() else 0)
            s = self._handle(Pn, An, qn[0], ln[0], un[0], device=dev_index)
            try:
                if world > 1:
                    table, xl, yl, (lo, hi) = sharded.solve_batch_sharded(s, q=qn, l=ln, u=un, rank=rank, world=world,
                                                                         device=device if q_val.is_cuda else None)
                    x = sharded.gather_rows(xl, nb, device=device if q_val.is_cuda else None)
                    rec = np.zeros((nb, 8)); rec[:, 0:5] = table[:, 1:6]
                else:
                    x, y, rec = s._solver.hip_batch_solve(q=qn, l=ln, u=un)
                    self.last_dual = y
            except ValueError as e:                                                    # only "does not fit one workgroup's LDS" falls back
                if str(e) != str(int(osqp_amd.SolverError.OSQP_FUNC_NOT_IMPLEMENTED)):
                    raise
                x, rec = self._loop(Pn, qn, An, ln, un, nb, batched, dev_index)
        else:
            dev_index = (device.index or 0) if q_val.is_cuda else (torch.cuda.current_device() if torch.cuda.is_available() else 0)
            x = rec = None
            if world == 1:                                                              # per-element matrices: one launch (osqp_hip_batch_solve_mat)
                # (the handle's OWN matrices matter only for the side that is shared; a batched side keeps what the handle holds -- no re-assembly per forward)
                have = self._solver is not None and self._device == dev_index
                s = self._handle((self._Pv if have else Pn[0]) if batched[0] else Pn, (self._Av if have else An[0]) if batched[2] else An, qn[0], ln[0], un[0], device=dev_index)
                try:
                    x, y, rec = s._solver.hip_batch_solve(q=qn, l=ln, u=un, Px=(Pn[:, self._triu_pick] if batched[0] else None), Ax=(An if batched[2] else None), nbatch=nb)
                    self.last_dual = y
                    self.mat_batch_launches = getattr(self, 'mat_batch_launches', 0) + 1
                except ValueError as e:
                    if str(e) != str(int(osqp_amd.SolverError.OSQP_FUNC_NOT_IMPLEMENTED)):
                        raise
                    x = rec = None
            if x is None:
                x, rec = self._loop(Pn, qn, An, ln, un, nb, batched, dev_index)
        bad = np.nonzero(rec[:, 0] != int(osqp_amd.SolverStatus.OSQP_SOLVED))[0]
        if bad.size:
            raise RuntimeError('Unable to solve QP, status: %d (batch element %d)' % (int(rec[bad[0], 0]), int(bad[0])))
        out = torch.as_tensor(x, dtype=dtype, device=device)
        return out if any(batched) else out.squeeze(0)

    def _forward_device(self, Pn, An, q_val, l_val, u_val, nb, rank=0, world=1):
        """q, l, u stay where they are (float64, contiguous, (nb, .) on q_val's device); x comes back as a device tensor.  In a
        distributed job (world > 1) this rank solves its contiguous share by device pointer and the shares meet in two all_gathers of
        device tensors (records, solutions) -- no host copy of q, l, u or x (sharded.solve_batch_sharded_device)."""
        dev = q_val.device
        exp = lambda t, k: t.detach().to(device=dev, dtype=torch.float64).expand(nb, k).contiguous()
        qd, ld, ud = exp(q_val, self.n), exp(l_val, self.m), exp(u_val, self.m)
        if self._solver is None or self._device != (dev.index or 0):                   # first forward only: the handle's own q, l, u
            s = self._handle(Pn, An, qd[0].cpu().numpy(), ld[0].cpu().numpy(), ud[0].cpu().numpy(), device=dev.index or 0)
        else:
            s = self._handle(Pn, An, None, None, None, device=dev.index or 0)
        if world > 1:
            try:
                table, xl, yl, (lo, hi) = sharded.solve_batch_sharded_device(s, q=qd, l=ld, u=ud, rank=rank, world=world)
            except ValueError as e:
                if str(e) != str(int(osqp_amd.SolverError.OSQP_FUNC_NOT_IMPLEMENTED)):
                    raise
                return None
            st = table[:, 1].to('cpu')                                                  # 8 bytes per problem: the statuses
            bad = torch.nonzero(st != int(osqp_amd.SolverStatus.OSQP_SOLVED)).flatten()
            if bad.numel():
                raise RuntimeError('Unable to solve QP, status: %d (batch element %d)' % (int(st[bad[0]]), int(bad[0])))
            self.last_dual = sharded.gather_rows_device(yl, nb)
            return sharded.gather_rows_device(xl, nb).to(q_val.dtype)
        x = torch.empty((nb, self.n), dtype=torch.float64, device=dev)
        y = torch.empty((nb, self.m), dtype=torch.float64, device=dev)
        rec = torch.empty((nb, 12), dtype=torch.float64, device=dev)    # OSQP_HIP_BATCH_REC
        stream = torch.cuda.current_stream(dev).cuda_stream
        try:
            s._solver.hip_batch_solve_device(nb, qd.data_ptr(), ld.data_ptr(), ud.data_ptr(), x.data_ptr(), y.data_ptr(), rec.data_ptr(),
                                             warm=False, stream=stream)
        except ValueError as e:
            if str(e) != str(int(osqp_amd.SolverError.OSQP_FUNC_NOT_IMPLEMENTED)):
                raise
            return None                                                                # does not fit one workgroup's LDS
        st = rec[:, 0].to('cpu')                                                        # (waits for the stream)
        bad = torch.nonzero(st != int(osqp_amd.SolverStatus.OSQP_SOLVED)).flatten()
        if bad.numel():
            raise RuntimeError('Unable to solve QP, status: %d (batch element %d)' % (int(st[bad[0]]), int(bad[0])))
        self.last_dual = y
        return x.to(q_val.dtype)

    def _loop(self, Pn, qn, An, ln, un, nb, batched, dev_index=0):
        """Per-element matrices (or a problem too large for the batch kernel): the single-QP engine, one element after the
        other on the persistent handle -- update(Px, Ax, q, l, u) + solve(), as nn/torch.py:136-157."""
        x = np.zeros((nb, self.n)); rec = np.zeros((nb, 8))
        for i in range(nb):
            Pv = Pn[i] if batched[0] else Pn
            Av = An[i] if batched[2] else An
            s = self._handle(Pv, Av, qn[i], ln[i], un[i], device=dev_index)
            s.update(q=qn[i], l=ln[i], u=un[i])
            r = s.solve()
            x[i] = r.x
            rec[i, 0], rec[i, 1], rec[i, 2] = r.info.status_val, r.info.iter, r.info.obj_val
        return x, rec
