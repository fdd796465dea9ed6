"""Forward pass of the reference's torch layer (/root/reference/src/osqp/nn/torch.py:22-230) on the MI355X engine.

The reference solves a batch of same-structure QPs with one osqp.OSQP object per batch element fanned out over
joblib threads (nn/torch.py:200-217).  Here:
  * P_val and A_val shared by the whole batch (1-D tensors)  -> ONE batched kernel launch for all elements
    (osqp_hip_batch_solve: one workgroup per problem; the reference's update(q,l,u)+solve() per element, :136-157);
    when q/l/u live on the GPU (torch ROCm tensors) they are handed over ZERO-COPY by device pointer
    (osqp_hip_batch_solve_device, enqueued on torch's current stream) and the solution tensor is produced on the device;
  * per-element P_val / A_val (2-D tensors) -> the single-QP engine, re-used through update(Px, Ax, q, l, u) (:136-140).
Like the reference, a batch element that is not solved raises RuntimeError (:158-162).
Backward (adjoint derivatives, :233-290) is out of scope of this engine: the returned tensor does not
carry a grad_fn.
"""
import numpy as np
import scipy.sparse as spa
import torch
from torch.nn import Module

import osqp_amd


def _np(t):
    return t.detach().cpu().double().numpy()


class OSQP(Module):
    def __init__(self, P_idx, P_shape, A_idx, A_shape, eps_rel=1e-5, eps_abs=1e-5, verbose=False, max_iter=10000, algebra='hip', solver_type='indirect'):
        super().__init__()
        self.P_idx, self.P_shape, self.A_idx, self.A_shape = P_idx, P_shape, A_idx, A_shape
        self.eps_rel, self.eps_abs, self.verbose, self.max_iter = eps_rel, eps_abs, verbose, max_iter
        self.algebra, self.solver_type = algebra, solver_type
        self.n, self.m = P_shape[0], A_shape[0]
        self._solver = None

    def _matrices(self, P_val, A_val):
        P = spa.csc_matrix((P_val, self.P_idx), shape=self.P_shape)
        A = spa.csc_matrix((A_val, self.A_idx), shape=self.A_shape)
        return P, A

    def _setup(self, P_val, q, A_val, l, u):
        P, A = self._matrices(P_val, A_val)
        s = osqp_amd.OSQP(algebra=self.algebra)
        s.setup(P, q, A, l, u, solver_type=self.solver_type, verbose=self.verbose, eps_abs=self.eps_abs, eps_rel=self.eps_rel,
                max_iter=self.max_iter, warm_starting=False)
        return s

    def forward(self, P_val, q_val, A_val, l_val, u_val):
        params = [P_val, q_val, A_val, l_val, u_val]
        dtype, device = q_val.dtype, q_val.device
        batched = [p.ndimension() == 2 for p in params]
        nb = max([p.size(0) for p, b in zip(params, batched) if b], default=1)
        Pn, qn, An, ln, un = (_np(p) for p in params)
        bc = lambda a, k: a if a.ndim == 2 else np.broadcast_to(a, (nb, k))          # nn/torch.py:184-188
        qn, ln, un = bc(qn, self.n), bc(ln, self.m), bc(un, self.m)
        if not batched[0] and not batched[2] and q_val.is_cuda:                      # shared matrices, data on the GPU: zero-copy
            out = self._forward_device(Pn, An, q_val, l_val, u_val, nb)
            if out is not None:
                return out if any(batched) else out.squeeze(0)
        if not batched[0] and not batched[2]:                                          # shared matrices: batched kernel
            self._solver = self._setup(Pn, qn[0], An, ln[0], un[0])
            try:
                x, y, rec = self._solver._solver.hip_batch_solve(q=qn, l=ln, u=un)
            except ValueError:                                                         # does not fit one workgroup's LDS
                x, rec = self._loop(Pn, qn, An, ln, un, nb, batched)
        else:
            x, rec = self._loop(Pn, qn, An, ln, un, nb, batched)
        bad = np.nonzero(rec[:, 0] != int(osqp_amd.SolverStatus.OSQP_SOLVED))[0]
        if bad.size:
            raise RuntimeError('Unable to solve QP, status: %d (batch element %d)' % (int(rec[bad[0], 0]), int(bad[0])))
        out = torch.as_tensor(x, dtype=dtype, device=device)
        return out if any(batched) else out.squeeze(0)

    def _forward_device(self, Pn, An, q_val, l_val, u_val, nb):
        """q, l, u stay where they are (float64, contiguous, (nb, .) on q_val's device); x comes back as a device tensor."""
        dev = q_val.device
        exp = lambda t, k: t.detach().to(device=dev, dtype=torch.float64).expand(nb, k).contiguous()
        qd, ld, ud = exp(q_val, self.n), exp(l_val, self.m), exp(u_val, self.m)
        s = osqp_amd.OSQP(algebra=self.algebra)
        s.setup(*self._matrices(Pn, An)[:1], qd[0].cpu().numpy(), self._matrices(Pn, An)[1], ld[0].cpu().numpy(), ud[0].cpu().numpy(),
                solver_type=self.solver_type, verbose=self.verbose, eps_abs=self.eps_abs, eps_rel=self.eps_rel, max_iter=self.max_iter,
                warm_starting=False, device=dev.index or 0)
        self._solver = s
        x = torch.empty((nb, self.n), dtype=torch.float64, device=dev)
        y = torch.empty((nb, self.m), dtype=torch.float64, device=dev)
        rec = torch.empty((nb, 8), dtype=torch.float64, device=dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        try:
            s._solver.hip_batch_solve_device(nb, qd.data_ptr(), ld.data_ptr(), ud.data_ptr(), x.data_ptr(), y.data_ptr(), rec.data_ptr(),
                                             warm=False, stream=stream)
        except ValueError:
            return None                                                                # does not fit one workgroup's LDS
        st = rec[:, 0].to('cpu')                                                        # (waits for the stream)
        bad = torch.nonzero(st != int(osqp_amd.SolverStatus.OSQP_SOLVED)).flatten()
        if bad.numel():
            raise RuntimeError('Unable to solve QP, status: %d (batch element %d)' % (int(st[bad[0]]), int(bad[0])))
        self.last_dual = y
        return x.to(q_val.dtype)

    def _loop(self, Pn, qn, An, ln, un, nb, batched):
        x = np.zeros((nb, self.n)); rec = np.zeros((nb, 8))
        s = None
        for i in range(nb):
            Pv = Pn[i] if batched[0] else Pn
            Av = An[i] if batched[2] else An
            if s is None:
                s = self._setup(Pv, qn[i], Av, ln[i], un[i])
                ptri = spa.triu(self._matrices(np.arange(1, len(Pv) + 1, dtype=float), Av)[0], format='csc')
                self._triu_pick = ptri.data.astype(int) - 1        # positions of the upper-triangle entries inside P_val
            else:
                s.update(q=qn[i], l=ln[i], u=un[i], Px=Pv[self._triu_pick], Ax=Av)
            r = s.solve()
            x[i] = r.x
            rec[i, 0], rec[i, 1], rec[i, 2] = r.info.status_val, r.info.iter, r.info.obj_val
        self._solver = s
        return x, rec
