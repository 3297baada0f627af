"""TEST INFRASTRUCTURE: a CPU stand-in for dist_vcycle.HipBackend so that the
row-block partition / halo-exchange / agglomeration logic of the multi-GPU
driver can run under torch.distributed's gloo backend without a GPU.

Arithmetic: the same per-row order as the HIP kernels and the oracle (ascending
column, separate multiply/add, IEEE divide), vectorised over rows with numpy.
The agglomerated coarse part is the CPU oracle's V-cycle.  Never imported by the
product."""
import numpy as np
import torch


class _Ell:
    def __init__(self, rowptr, col, val):
        n = rowptr.size - 1
        cnt = np.diff(rowptr)
        w = int(cnt.max()) if n else 0
        self.n, self.w = n, w
        self.col = np.zeros((n, max(w, 1)), np.int64)
        self.val = np.zeros((n, max(w, 1)), np.float64)
        self.mask = np.zeros((n, max(w, 1)), bool)
        for j in range(w):
            live = cnt > j
            idx = rowptr[:-1][live] + j
            self.col[live, j] = col[idx]
            self.val[live, j] = val[idx]
            self.mask[live, j] = True


class CpuBackend:
    device = torch.device("cpu")

    def __init__(self, oracle):
        self.O = oracle

    def vec(self, n):
        return torch.zeros(n, dtype=torch.float64)

    def from_numpy(self, a):
        return torch.from_numpy(np.ascontiguousarray(a))

    def to_numpy(self, t):
        return t.detach().numpy()

    def matrix(self, rowptr, col, val, ncols=None, diag_shift=0):
        return _Ell(rowptr, col, val)

    def residual(self, m, u_ext, f, r):
        u = u_ext.numpy()
        acc = f.numpy().copy()
        for j in range(m.w):
            acc = np.where(m.mask[:, j], acc - m.val[:, j] * u[m.col[:, j]], acc)
        r.copy_(torch.from_numpy(acc))

    def jacobi(self, m, u_ext, b, u_out, omega, diag_shift):
        u = u_ext.numpy()
        rows = np.arange(m.n) + diag_shift
        acc = np.zeros(m.n)
        diag = np.zeros(m.n)
        for j in range(m.w):
            is_d = m.mask[:, j] & (m.col[:, j] == rows)
            off = m.mask[:, j] & ~is_d
            diag = np.where(is_d, m.val[:, j], diag)
            acc = np.where(off, acc + m.val[:, j] * u[m.col[:, j]], acc)
        uk = u[rows]
        with np.errstate(divide="ignore", invalid="ignore"):
            new = uk + omega * ((b.numpy() - acc) / diag - uk)
        u_out.copy_(torch.from_numpy(np.where(diag == 0.0, uk, new)))

    def jacobi_from_zero(self, diag, b, u_out, omega):
        d = diag.numpy()
        with np.errstate(divide="ignore", invalid="ignore"):
            new = 0.0 + omega * ((b.numpy() - 0.0) / d - 0.0)
        u_out.copy_(torch.from_numpy(np.where(d == 0.0, 0.0, new)))

    def spmv(self, m, v_ext, out):
        v = v_ext.numpy()
        acc = np.zeros(m.n)
        for j in range(m.w):
            acc = np.where(m.mask[:, j], acc + m.val[:, j] * v[m.col[:, j]], acc)
        out.copy_(torch.from_numpy(acc))

    def add_(self, y, x):
        y.copy_(y + x)

    def sumsq(self, r):
        return torch.tensor([float(np.sum(r.numpy() ** 2))], dtype=torch.float64)

    def sync(self):
        pass

    def tail(self, colptr, rowind, val, n_levels, omega, sweeps):
        return _OracleTail(self.O, colptr, rowind, val, n_levels, omega, sweeps)


class _OracleTail:
    def __init__(self, O, colptr, rowind, val, n_levels, omega, sweeps):
        n = colptr.size - 1
        self.O = O
        self.A = O.CSC(n, n, colptr, rowind, val)
        self.mg = O.Multigrid(self.A, np.zeros(n), n_levels, smoother=O.SM_TRUE_JACOBI,
                              smoother_iters=sweeps, omega=omega)

    def cycle(self, f_full, u_full, zero_guess=True):
        self.mg.set_vec(0, "f", f_full.numpy())
        if zero_guess:
            self.mg.set_vec(0, "u", np.zeros(f_full.numel()))
        self.mg.vcycle()
        u_full.copy_(torch.from_numpy(self.mg.get_vec(0, "u")))

    def rss(self):
        return self.mg.rss()
