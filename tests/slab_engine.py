"""TEST INFRASTRUCTURE: a CPU stand-in for slab_vcycle.HipSlabEngine, so that the slab-sharded
V-cycle (algebraic-multigrid_amd/slab_vcycle.py: halo exchange, all-gather, the line ranges of
amg_hip_slab_plan) runs under gloo without a GPU.

It executes the three parts of amg_hip_slab_run the way the K-Patch kernels do -- each leg only
over the plan's line range of its level -- with the oracle's per-row arithmetic, and POISONS
with NaN every entry the plan does not promise to be valid (everything outside a leg's range,
and the level-0 solution outside the owned lines + halo).  A halo that is one line too shallow
therefore shows up as NaN in the assembled solution instead of depending on the data.
Never imported by the product."""
import numpy as np
import torch

from cpu_backend import _Ell


def _jacobi(E, u, f, omega, r0, r1):
    sl = slice(r0, r1)
    rows = np.arange(r0, r1)
    acc = np.zeros(r1 - r0)
    diag = np.zeros(r1 - r0)
    for j in range(E.w):
        mk, cj, vj = E.mask[sl, j], E.col[sl, j], E.val[sl, j]
        is_d = mk & (cj == rows)
        off = mk & ~is_d
        diag = np.where(is_d, vj, diag)
        acc = np.where(off, acc + vj * u[cj], acc)
    uk = u[sl]
    with np.errstate(divide="ignore", invalid="ignore"):
        new = uk + omega * ((f[sl] - acc) / diag - uk)
    return np.where(diag == 0.0, uk, new)


def _residual(E, u, f, r0, r1):
    sl = slice(r0, r1)
    acc = f[sl].copy()
    for j in range(E.w):
        acc = np.where(E.mask[sl, j], acc - E.val[sl, j] * u[E.col[sl, j]], acc)
    return acc


def _restrict(r, n, c0, c1):
    """f_H[c] for c in [c0, c1): ((0.5 r[2c]) + 1.0 r[2c+1]) + 0.5 r[2c+2] (interpolator.hpp:64-68)"""
    c = np.arange(c0, c1)
    i = 2 * c
    s = np.zeros(c1 - c0)
    s = np.where(i < n, s + 0.5 * r[np.minimum(i, n - 1)], s)
    s = np.where(i + 1 < n, s + 1.0 * r[np.minimum(i + 1, n - 1)], s)
    s = np.where(i + 2 < n, s + 0.5 * r[np.minimum(i + 2, n - 1)], s)
    return s


def _prolong_add(x, uH, nH, r0, r1):
    """(x + P uH)[r0:r1] (interpolator.hpp:52-56, multigrid.hpp:294-296)"""
    i = np.arange(r0, r1)
    j = i >> 1
    odd = (i & 1) != 0
    a_ok = ~odd & (j >= 1) & (j - 1 < nH)
    b_ok = j < nH
    a = uH[np.where(a_ok, j - 1, 0)]
    b = uH[np.where(b_ok, j, 0)]
    t = np.zeros(r1 - r0)
    t = np.where(a_ok, t + 0.5 * a, t)
    t = np.where(b_ok, t + np.where(odd, 1.0, 0.5) * b, t)
    return x[r0:r1] + t


class EmulatedSlabEngine:
    """n x n 5-point Poisson (n a power of two), L levels, true Jacobi 2+2; `levels` slab levels."""

    def __init__(self, O, amg, n, L, omega, rank, world, levels, tamper=0):
        self.O, self.omega, self.rank, self.world = O, omega, rank, world
        A, b = O.laplacian(n), O.rhs(n)
        self.ref = O.Multigrid(A, b, L, smoother=O.SM_TRUE_JACOBI, smoother_iters=2, omega=omega)
        self.L = L
        self.n = [self.ref.n_dofs(l) for l in range(L)]
        self.mats = [self.ref.level_matrix(l) for l in range(L)]
        self.E = [_Ell(M.colptr, M.rowind, M.val) for M in self.mats[:L - 1]]   # symmetric: CSC == CSR
        self.diag = []
        for l in range(L):
            M = self.mats[l]
            cols = np.repeat(np.arange(self.n[l]), np.diff(M.colptr))
            d = np.zeros(self.n[l])
            on = M.rowind == cols
            d[cols[on]] = M.val[on]
            self.diag.append(d)
        info = amg.slab_plan(n, rank, world, levels)
        info.pitch0 = n
        info.gather_pitch = n >> levels
        info.gather_rows = self.n[levels]
        if tamper:   # a halo `tamper` lines too shallow on the inner sides (sensitivity check)
            for l in range(levels):
                if rank > 0:
                    info.down_lo[l] += tamper
                if rank < world - 1:
                    info.down_hi[l] -= tamper
            info.halo_lines -= tamper
        self.info = info
        self.k = levels
        cap = int(info.chunk_lines) * world
        self.u0 = torch.zeros(cap * n, dtype=torch.float64)
        self.fg = torch.zeros(cap * (n >> levels), dtype=torch.float64)
        self.f = [b.copy()] + [np.zeros(self.n[l]) for l in range(1, L)]
        # per level: `first` = result of the from-zero sweep / final u of the level (Level::tmp in
        # solver.cpp for l >= 1), `sm` = the smoothed u between the two legs
        self.first = [None] * L
        self.sm = [None] * L
        self.u = [np.zeros(self.n[l]) for l in range(L)]

    def _rows(self, l, lo, hi):
        m = int(self.info.pitch0) >> l
        return min(lo * m, self.n[l]), min(hi * m, self.n[l])

    def _nan(self, l):
        return np.full(self.n[l], np.nan)

    def run(self, part):
        i, k, w = self.info, self.k, self.omega
        if part == 1:
            u0 = self.u0.numpy()[:self.n[0]]
            if self.world > 1:   # only the owned lines and the exchanged halo are promised
                a, b = self._rows(0, max(0, int(i.line_begin) - int(i.halo_lines)),
                                  min(int(i.lines), int(i.line_end) + int(i.halo_lines)))
                u0[:a] = np.nan
                u0[b:] = np.nan
            for l in range(k):
                E, f, n = self.E[l], self.f[l], self.n[l]
                r0, r1 = self._rows(l, int(i.down_lo[l]), int(i.down_hi[l]))
                x = u0 if l == 0 else self.first[l]
                if l == 0:   # both pre-sweeps; below level 0 the first one came from the finer leg
                    s1 = self._nan(l)
                    s1[r0:r1] = _jacobi(E, x, f, w, r0, r1)
                    x = s1
                sm = self._nan(l)
                sm[r0:r1] = _jacobi(E, x, f, w, r0, r1)
                self.sm[l] = sm
                r = self._nan(l)
                r[r0:r1] = _residual(E, sm, f, r0, r1)
                nH = self.n[l + 1]
                c0, c1 = min((r0 + 1) // 2, nH), min((r1 + 1) // 2, nH)   # coarse rows c with r0 <= 2c < r1
                fH = self.fg.numpy()[:nH] if l + 1 == k else self._nan(l + 1)
                if l + 1 == k and self.world > 1:
                    fH[:] = np.nan
                fH[c0:c1] = _restrict(r, n, c0, c1)
                d = self.diag[l + 1][c0:c1]
                first = self._nan(l + 1)
                with np.errstate(divide="ignore", invalid="ignore"):
                    first[c0:c1] = np.where(d == 0.0, 0.0, 0.0 + w * ((fH[c0:c1] - 0.0) / d - 0.0))
                self.f[l + 1] = fH
                self.first[l + 1] = first
        elif part == 2:
            self._tail()
        else:
            for l in range(k - 1, -1, -1):
                E, f, n = self.E[l], self.f[l], self.n[l]
                r0, r1 = self._rows(l, int(i.up_lo[l]), int(i.up_hi[l]))
                uH = self.first[l + 1]          # the coarser level's final u
                # patch_up_kernel: loads three lines beyond the tile, first sweep on one line and
                # one entry beyond, second sweep on the tile
                e0, e1 = self._rows(l, max(0, int(i.up_lo[l]) - 3), min(int(i.lines), int(i.up_hi[l]) + 3))
                x = self._nan(l)
                x[e0:e1] = _prolong_add(self.sm[l], uH, self.n[l + 1], e0, e1)
                g0, g1 = self._rows(l, max(0, int(i.up_lo[l]) - 1), min(int(i.lines), int(i.up_hi[l]) + 1))
                g0, g1 = max(0, g0 - 1), min(n, g1 + 1)
                s1 = self._nan(l)
                s1[g0:g1] = _jacobi(E, x, f, w, g0, g1)
                if l == 0:
                    self.u0.numpy()[r0:r1] = _jacobi(E, s1, f, w, r0, r1)
                else:
                    out = self._nan(l)
                    out[r0:r1] = _jacobi(E, s1, f, w, r0, r1)
                    self.first[l] = out

    def _tail(self):
        """levels k .. L-1 on whole vectors, replicated (the rest of multigrid.hpp:263-305)"""
        k, L, w = self.k, self.L, self.omega
        fk = self.fg.numpy()[:self.n[k]].copy()
        self.f[k] = fk
        f = self.f
        u = [None] * L
        for l in range(k, L - 1):
            n = self.n[l]
            d = self.diag[l]
            with np.errstate(divide="ignore", invalid="ignore"):
                x = np.where(d == 0.0, 0.0, 0.0 + w * ((f[l] - 0.0) / d - 0.0))   # from-zero sweep
            x = _jacobi(self.E[l], x, f[l], w, 0, n)
            u[l] = x
            r = _residual(self.E[l], x, f[l], 0, n)
            f[l + 1] = _restrict(r, n, 0, self.n[l + 1])
        u[L - 1] = self.O.band_solve(self.mats[L - 1], f[L - 1])[0]
        for l in range(L - 2, k - 1, -1):
            n = self.n[l]
            x = _prolong_add(u[l], u[l + 1], self.n[l + 1], 0, n)
            x = _jacobi(self.E[l], x, f[l], w, 0, n)
            u[l] = _jacobi(self.E[l], x, f[l], w, 0, n)
        self.first[k] = u[k]

    def rss(self):
        return self.O.rss(self.mats[0], self.u0.numpy()[:self.n[0]].copy(), self.f[0])

    def sync(self):
        pass

    def close(self):
        pass
