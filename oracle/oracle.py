"""ctypes binding of the CPU oracle (oracle/amg_oracle.cpp).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libamg_oracle.so")

SM_SPGS, SM_REF_JACOBI, SM_SOR, SM_TRUE_JACOBI, SM_MULTICOLOR = 0, 1, 2, 3, 4


def build(force=False):
    src = os.path.join(_HERE, "amg_oracle.cpp")
    if (force or not os.path.exists(_LIB_PATH)
            or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src)):
        subprocess.check_call(["make", "-C", _HERE, "-B"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None

_i32p = C.POINTER(C.c_int32)
_f64p = C.POINTER(C.c_double)


def _p32(a):
    return None if a is None else a.ctypes.data_as(_i32p)


def _p64(a):
    return None if a is None else a.ctypes.data_as(_f64p)


def _bind(L):
    L.orc_grid_spacing_h.restype = C.c_double
    L.orc_grid_spacing_h.argtypes = [C.c_uint64]
    L.orc_points_n_from_grid_spacing_h.restype = C.c_uint64
    L.orc_points_n_from_grid_spacing_h.argtypes = [C.c_double]
    L.orc_laplacian.restype = C.c_int64
    L.orc_laplacian.argtypes = [C.c_int, C.c_uint64, _i32p, _i32p, _f64p]
    L.orc_rhs.restype = None
    L.orc_rhs.argtypes = [C.c_int, C.c_uint64, _f64p]
    L.orc_n_H_from_n_h.restype = C.c_uint64
    L.orc_n_H_from_n_h.argtypes = [C.c_uint64]
    L.orc_make_P.restype = C.c_int64
    L.orc_make_P.argtypes = [C.c_uint64, C.c_uint64, _i32p, _i32p, _f64p]
    L.orc_transpose.restype = None
    L.orc_transpose.argtypes = [C.c_int64, C.c_int64, _i32p, _i32p, _f64p,
                                _i32p, _i32p, _f64p]
    L.orc_residual.restype = None
    L.orc_residual.argtypes = [C.c_int64, _i32p, _i32p, _f64p, _f64p, _f64p, _f64p]
    L.orc_spmv.restype = None
    L.orc_spmv.argtypes = [C.c_int64, C.c_int64, _i32p, _i32p, _f64p, _f64p, _f64p]
    L.orc_rss.restype = C.c_double
    L.orc_rss.argtypes = [C.c_int64, _i32p, _i32p, _f64p, _f64p, _f64p]
    L.orc_smooth.restype = C.c_uint64
    L.orc_smooth.argtypes = [C.c_int, C.c_int64, _i32p, _i32p, _f64p, _f64p, _f64p,
                             C.c_double, C.c_double, C.c_uint64, C.c_uint64,
                             _i32p, C.c_int32, C.POINTER(C.c_int)]
    L.orc_spgs_sweep.restype = None
    L.orc_spgs_sweep.argtypes = [C.c_int, C.c_int64, _i32p, _i32p, _f64p, _f64p, _f64p]
    L.orc_band_solve.restype = C.c_int64
    L.orc_band_solve.argtypes = [C.c_int64, _i32p, _i32p, _f64p, _f64p, _f64p]
    L.orc_mg_create.restype = C.c_void_p
    L.orc_mg_create.argtypes = [C.c_int64, _i32p, _i32p, _f64p, _f64p, C.c_uint64]
    L.orc_mg_create_custom.restype = C.c_void_p
    L.orc_mg_create_custom.argtypes = [C.c_int64, _i32p, _i32p, _f64p, _f64p, C.c_uint64,
                                       C.POINTER(C.c_int64), C.POINTER(_i32p), C.POINTER(_i32p),
                                       C.POINTER(_f64p)]
    L.orc_spgemm.restype = C.c_int64
    L.orc_spgemm.argtypes = [C.c_int64, C.c_int64, _i32p, _i32p, _f64p, C.c_int64, _i32p, _i32p, _f64p,
                             _i32p, _i32p, _f64p]
    L.orc_mg_destroy.restype = None
    L.orc_mg_destroy.argtypes = [C.c_void_p]
    L.orc_mg_set_smoother.restype = None
    L.orc_mg_set_smoother.argtypes = [C.c_void_p, C.c_int, C.c_uint64, C.c_double]
    L.orc_mg_set_colors.restype = None
    L.orc_mg_set_colors.argtypes = [C.c_void_p, C.c_uint64, _i32p, C.c_int32]
    L.orc_mg_n_dofs.restype = C.c_uint64
    L.orc_mg_n_dofs.argtypes = [C.c_void_p, C.c_uint64]
    L.orc_mg_level_nnz.restype = C.c_int64
    L.orc_mg_level_nnz.argtypes = [C.c_void_p, C.c_uint64]
    L.orc_mg_level_matrix.restype = None
    L.orc_mg_level_matrix.argtypes = [C.c_void_p, C.c_uint64, _i32p, _i32p, _f64p]
    L.orc_mg_transfer_nnz.restype = C.c_int64
    L.orc_mg_transfer_nnz.argtypes = [C.c_void_p, C.c_uint64, C.c_int]
    L.orc_mg_transfer.restype = None
    L.orc_mg_transfer.argtypes = [C.c_void_p, C.c_uint64, C.c_int, _i32p, _i32p, _f64p]
    L.orc_mg_get_vec.restype = None
    L.orc_mg_get_vec.argtypes = [C.c_void_p, C.c_uint64, C.c_int, _f64p]
    L.orc_mg_set_vec.restype = None
    L.orc_mg_set_vec.argtypes = [C.c_void_p, C.c_uint64, C.c_int, _f64p]
    L.orc_mg_coarse_halfbw.restype = C.c_int64
    L.orc_mg_coarse_halfbw.argtypes = [C.c_void_p]
    L.orc_mg_vcycle.restype = None
    L.orc_mg_vcycle.argtypes = [C.c_void_p]
    L.orc_mg_rss.restype = C.c_double
    L.orc_mg_rss.argtypes = [C.c_void_p]
    L.orc_mg_solve.restype = C.c_uint64
    L.orc_mg_solve.argtypes = [C.c_void_p, C.c_double, C.c_uint64, C.c_uint64,
                               C.POINTER(C.c_int), _f64p, _f64p, C.c_uint64]
    L.orc_mg_apply.restype = None
    L.orc_mg_apply.argtypes = [C.c_void_p, _f64p, _f64p]
    L.orc_mg_pcg.restype = C.c_uint64
    L.orc_mg_pcg.argtypes = [C.c_void_p, C.c_double, C.c_uint64, _f64p]
    L.orc_mg_time_vcycles.restype = C.c_double
    L.orc_mg_time_vcycles.argtypes = [C.c_void_p, C.c_uint64]
    return L


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = _bind(C.CDLL(_LIB_PATH))
    return _lib


_variants = {}


def lib_variant(flags):
    """The same restatement compiled with other optimisation flags (bench.py times the
    CPU baseline both as -O2 and as -O3 -march=native).  Built on the machine
    that runs it (-march=native must not travel), under oracle/_build/."""
    import hashlib
    import platform
    if flags in _variants:
        return _variants[flags]
    try:
        cpu = open("/proc/cpuinfo").read().split("flags", 1)[1].split("\n", 1)[0]
    except Exception:
        cpu = platform.processor()
    src = os.path.join(_HERE, "amg_oracle.cpp")
    tag = hashlib.sha256((flags + cpu + open(src).read()).encode()).hexdigest()[:12]
    out_dir = os.path.join(_HERE, "_build")
    os.makedirs(out_dir, exist_ok=True)
    path = os.path.join(out_dir, f"libamg_oracle_{tag}.so")
    if not os.path.exists(path):
        tmp = path + f".{os.getpid()}.tmp"
        subprocess.check_call(["g++"] + flags.split() + ["-fPIC", "-std=c++17", "-shared", "-o", tmp, src])
        os.replace(tmp, path)
    _variants[flags] = _bind(C.CDLL(path))
    return _variants[flags]


class CSC:
    """Column-major sparse matrix, int32 indices (Eigen::SparseMatrix<double>)."""

    def __init__(self, rows, cols, colptr, rowind, val):
        self.rows, self.cols = int(rows), int(cols)
        self.colptr = np.ascontiguousarray(colptr, dtype=np.int32)
        self.rowind = np.ascontiguousarray(rowind, dtype=np.int32)
        self.val = np.ascontiguousarray(val, dtype=np.float64)

    @property
    def nnz(self):
        return int(self.rowind.size)

    def to_scipy(self):
        import scipy.sparse as sp
        return sp.csc_matrix((self.val, self.rowind, self.colptr),
                             shape=(self.rows, self.cols))

    def transpose(self):
        tc = np.empty(self.rows + 1, np.int32)
        tr = np.empty(self.nnz, np.int32)
        tv = np.empty(self.nnz, np.float64)
        lib().orc_transpose(self.rows, self.cols, _p32(self.colptr), _p32(self.rowind),
                            _p64(self.val), _p32(tc), _p32(tr), _p64(tv))
        return CSC(self.cols, self.rows, tc, tr, tv)


def laplacian(n, dim=2):
    """grid.hpp:88-98 (dim=2); dim=3 is the build-side 7-point analogue."""
    N = n ** dim
    nnz = lib().orc_laplacian(dim, n, None, None, None)
    colptr = np.empty(N + 1, np.int32)
    rowind = np.empty(nnz, np.int32)
    val = np.empty(nnz, np.float64)
    lib().orc_laplacian(dim, n, _p32(colptr), _p32(rowind), _p64(val))
    return CSC(N, N, colptr, rowind, val)


def rhs(n, dim=2):
    """grid.hpp:108-140 with the default Gaussian forcing."""
    b = np.empty(n ** dim, np.float64)
    lib().orc_rhs(dim, n, _p64(b))
    return b


def grid_spacing_h(n):
    return lib().orc_grid_spacing_h(n)


def points_n_from_grid_spacing_h(h):
    return int(lib().orc_points_n_from_grid_spacing_h(h))


def n_H_from_n_h(n_h):
    return int(lib().orc_n_H_from_n_h(n_h))


def make_P(n_h, n_H):
    nnz = lib().orc_make_P(n_h, n_H, None, None, None)
    colptr = np.empty(n_H + 1, np.int32)
    rowind = np.empty(nnz, np.int32)
    val = np.empty(nnz, np.float64)
    lib().orc_make_P(n_h, n_H, _p32(colptr), _p32(rowind), _p64(val))
    return CSC(n_h, n_H, colptr, rowind, val)


def residual(A, u, f):
    r = np.empty(A.rows, np.float64)
    lib().orc_residual(A.rows, _p32(A.colptr), _p32(A.rowind), _p64(A.val),
                       _p64(np.ascontiguousarray(u)), _p64(np.ascontiguousarray(f)), _p64(r))
    return r


def spmv(M, v):
    out = np.empty(M.rows, np.float64)
    lib().orc_spmv(M.rows, M.cols, _p32(M.colptr), _p32(M.rowind), _p64(M.val),
                   _p64(np.ascontiguousarray(v)), _p64(out))
    return out


def rss(A, u, b):
    return lib().orc_rss(A.rows, _p32(A.colptr), _p32(A.rowind), _p64(A.val),
                         _p64(np.ascontiguousarray(u)), _p64(np.ascontiguousarray(b)))


def smooth(kind, A, u, b, n_iters=1, omega=1.0, tol=1e-9, every=0, color=None,
           n_colors=0):
    """Runs smoother `kind` in place on a copy of u; returns (u, iters, converged)."""
    u = np.array(u, dtype=np.float64, copy=True)
    conv = C.c_int(0)
    col = None if color is None else np.ascontiguousarray(color, dtype=np.int32)
    it = lib().orc_smooth(kind, A.rows, _p32(A.colptr), _p32(A.rowind), _p64(A.val),
                          _p64(u), _p64(np.ascontiguousarray(b)), omega, tol, every,
                          n_iters, _p32(col), n_colors, C.byref(conv))
    return u, int(it), bool(conv.value)


def spgs_sweep(direction, A, u, b):
    u = np.array(u, dtype=np.float64, copy=True)
    lib().orc_spgs_sweep(direction, A.rows, _p32(A.colptr), _p32(A.rowind),
                         _p64(A.val), _p64(u), _p64(np.ascontiguousarray(b)))
    return u


def band_solve(A, f):
    x = np.empty(A.rows, np.float64)
    w = lib().orc_band_solve(A.rows, _p32(A.colptr), _p32(A.rowind), _p64(A.val),
                             _p64(np.ascontiguousarray(f)), _p64(x))
    return x, int(w)


def spgemm(A, B):
    """C = A B, Eigen's conservative sparse product (what multigrid.hpp:222 runs)."""
    nnz = lib().orc_spgemm(A.rows, A.cols, _p32(A.colptr), _p32(A.rowind), _p64(A.val), B.cols,
                           _p32(B.colptr), _p32(B.rowind), _p64(B.val), None, None, None)
    cp, ri, v = np.empty(B.cols + 1, np.int32), np.empty(nnz, np.int32), np.empty(nnz, np.float64)
    lib().orc_spgemm(A.rows, A.cols, _p32(A.colptr), _p32(A.rowind), _p64(A.val), B.cols,
                     _p32(B.colptr), _p32(B.rowind), _p64(B.val), _p32(cp), _p32(ri), _p64(v))
    return CSC(A.rows, B.cols, cp, ri, v)


def ruge_stueben_P(A, theta=0.25):
    """Strength-based C/F splitting + direct interpolation: the twin of the product's
    host_setup.cpp: ruge_stueben_P (NO counterpart in the reference, whose README.md:104-109
    only names the method).  Written from the textbook description (Briggs, Henson,
    McCormick: A Multigrid Tutorial, ch. 8; Ruge & Stueben 1987), plain Python on purpose:
      s_ij = -sign(a_ii) a_ij;  i depends strongly on j  <=>  s_ij > 0 and s_ij >= theta max_k s_ik
      first pass: the undecided point with the most strong dependants (ties: lowest index)
        becomes C, its undecided dependants F, the undecided points those depend on gain one,
        the undecided points the new C point depends on lose one; no strong coupling -> F
      F row i over C_i = strong C neighbours (ascending column sums):
        w_ij = -(sum_{s_ik>0} a_ik / sum_{k in C_i} a_ik) a_ij / (a_ii + sum_{s_ik<0} a_ik)
    Returns (P as CSC n x n_c, is_c)."""
    import heapq
    R = A.transpose()           # CSC(A^T) = rows of A
    n = A.rows
    ptr, idx, val = R.colptr, R.rowind, R.val
    S = [[] for _ in range(n)]
    ST = [[] for _ in range(n)]
    sgn = np.ones(n)
    for i in range(n):
        d = 0.0
        for p in range(ptr[i], ptr[i + 1]):
            if idx[p] == i:
                d = val[p]
        sgn[i] = -1.0 if d < 0.0 else 1.0
        mx = 0.0
        for p in range(ptr[i], ptr[i + 1]):
            if idx[p] != i:
                mx = max(mx, -sgn[i] * val[p])
        if mx > 0.0:
            thr = theta * mx
            for p in range(ptr[i], ptr[i + 1]):
                sij = -sgn[i] * val[p]
                if idx[p] != i and sij > 0.0 and sij >= thr:
                    S[i].append(int(idx[p]))
    for i in range(n):
        for j in S[i]:
            ST[j].append(i)
    U, Cp, Fp = 0, 1, 2
    st = [U] * n
    lam = [len(ST[i]) for i in range(n)]
    heap = []
    for i in range(n):
        if not S[i]:
            st[i] = Fp
        else:
            heap.append((-lam[i], i))
    heapq.heapify(heap)
    while heap:
        ml, i = heapq.heappop(heap)
        if st[i] != U or lam[i] != -ml:
            continue
        st[i] = Cp
        for j in ST[i]:
            if st[j] != U:
                continue
            st[j] = Fp
            for k in S[j]:
                if st[k] == U:
                    lam[k] += 1
                    heapq.heappush(heap, (-lam[k], k))
        for j in S[i]:
            if st[j] == U:
                lam[j] -= 1
                heapq.heappush(heap, (-lam[j], j))
    cidx = [-1] * n
    nc = 0
    for i in range(n):
        if st[i] == Cp:
            cidx[i] = nc
            nc += 1
    rows_ptr, rows_idx, rows_val = [0], [], []
    for i in range(n):
        if st[i] == Cp:
            rows_idx.append(cidx[i])
            rows_val.append(1.0)
        else:
            num = den = dg = 0.0
            a = {}
            for p in range(ptr[i], ptr[i + 1]):
                if idx[p] == i:
                    dg = float(val[p])
            for p in range(ptr[i], ptr[i + 1]):
                j = int(idx[p])
                if j == i:
                    continue
                a[j] = float(val[p])
                sij = -sgn[i] * val[p]
                if sij > 0.0:
                    num += float(val[p])
                elif sij < 0.0:
                    dg += float(val[p])
            for j in S[i]:
                if st[j] == Cp:
                    den += a[j]
            if den != 0.0 and dg != 0.0:
                alpha = num / den
                for j in S[i]:
                    if st[j] == Cp:
                        rows_idx.append(cidx[j])
                        rows_val.append(-alpha * a[j] / dg)
        rows_ptr.append(len(rows_idx))
    Pr = CSC(nc, n, rows_ptr, rows_idx, rows_val)      # CSC(P^T) = rows of P
    return Pr.transpose(), np.array([s == Cp for s in st])


def ruge_stueben_hierarchy(A, max_levels=25, theta=0.25, min_coarse=500):
    """P_l for amg_hip_create_rs's rule: stop at max_levels, at a level of at most min_coarse
    rows, or where a level no longer coarsens; A_{l+1} = R (A P) with R = P^T (multigrid.hpp:219-223)."""
    Ps = []
    Al = A
    while len(Ps) + 1 < max_levels and Al.rows > min_coarse:
        P, _ = ruge_stueben_P(Al, theta)
        if P.cols < 1 or P.cols >= Al.rows:
            break
        Ps.append(P)
        Al = spgemm(P.transpose(), spgemm(Al, P))
    return Ps


class Multigrid:
    """Restatement of AMG::Multigrid<double> (multigrid.hpp) with
    LinearInterpolator and a selectable smoother (default SparseGaussSeidel())."""

    def __init__(self, A, b, n_levels, smoother=SM_SPGS, smoother_iters=1, omega=1.0, library=None,
                 transfers=None):
        """transfers: P_l (CSC objects, n_l x n_{l+1}) of a custom interpolator; R_l = P_l^T."""
        self._L = library or lib()
        if transfers is None:
            self._h = self._L.orc_mg_create(A.rows, _p32(A.colptr), _p32(A.rowind),
                                            _p64(A.val), _p64(np.ascontiguousarray(b)), n_levels)
        else:
            assert len(transfers) == n_levels - 1
            k = max(len(transfers), 1)
            cols = (C.c_int64 * k)(*[P.cols for P in transfers])
            cp = (_i32p * k)(*[_p32(P.colptr) for P in transfers])
            ri = (_i32p * k)(*[_p32(P.rowind) for P in transfers])
            vv = (_f64p * k)(*[_p64(P.val) for P in transfers])
            self._h = self._L.orc_mg_create_custom(A.rows, _p32(A.colptr), _p32(A.rowind), _p64(A.val),
                                                   _p64(np.ascontiguousarray(b)), n_levels, cols, cp, ri, vv)
        if not self._h:
            raise ValueError("orc_mg_create failed")
        self.n_levels = n_levels
        self._L.orc_mg_set_smoother(self._h, smoother, smoother_iters, omega)

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.orc_mg_destroy(self._h)
            self._h = None

    def set_colors(self, level, color, n_colors):
        c = np.ascontiguousarray(color, dtype=np.int32)
        self._L.orc_mg_set_colors(self._h, level, _p32(c), n_colors)

    def n_dofs(self, level):
        return int(self._L.orc_mg_n_dofs(self._h, level))

    def level_matrix(self, level):
        n = self.n_dofs(level)
        nnz = self._L.orc_mg_level_nnz(self._h, level)
        colptr = np.empty(n + 1, np.int32)
        rowind = np.empty(nnz, np.int32)
        val = np.empty(nnz, np.float64)
        self._L.orc_mg_level_matrix(self._h, level, _p32(colptr), _p32(rowind), _p64(val))
        return CSC(n, n, colptr, rowind, val)

    def transfer(self, level, which):
        """which: 'P' (n_h x n_H) or 'R' (n_H x n_h) between level and level+1."""
        w = 1 if which == "R" else 0
        n_h, n_H = self.n_dofs(level), self.n_dofs(level + 1)
        rows, cols = (n_H, n_h) if w else (n_h, n_H)
        nnz = self._L.orc_mg_transfer_nnz(self._h, level, w)
        colptr = np.empty(cols + 1, np.int32)
        rowind = np.empty(nnz, np.int32)
        val = np.empty(nnz, np.float64)
        self._L.orc_mg_transfer(self._h, level, w, _p32(colptr), _p32(rowind), _p64(val))
        return CSC(rows, cols, colptr, rowind, val)

    def get_vec(self, level, which):
        out = np.empty(self.n_dofs(level), np.float64)
        self._L.orc_mg_get_vec(self._h, level, {"u": 0, "f": 1, "r": 2}[which], _p64(out))
        return out

    def set_vec(self, level, which, v):
        v = np.ascontiguousarray(v, dtype=np.float64)
        assert v.size == self.n_dofs(level)
        self._L.orc_mg_set_vec(self._h, level, {"u": 0, "f": 1, "r": 2}[which], _p64(v))

    def coarse_halfbw(self):
        return int(self._L.orc_mg_coarse_halfbw(self._h))

    def vcycle(self):
        self._L.orc_mg_vcycle(self._h)

    def rss(self):
        return self._L.orc_mg_rss(self._h)

    def solve(self, tol=1e-9, every=10, n_iters=100):
        """Returns (iters, converged, last_rss, trajectory-of-rss-checks)."""
        conv = C.c_int(0)
        last = C.c_double(0)
        cap = n_iters // max(every, 1) + 1
        traj = np.zeros(cap, np.float64)
        it = self._L.orc_mg_solve(self._h, tol, every, n_iters, C.byref(conv),
                                C.byref(last), _p64(traj), cap)
        k = int(it) // every
        return int(it), bool(conv.value), last.value, traj[:k].copy()

    def apply(self, v):
        """z = M^-1 v: one V-cycle from zero with v as the right-hand side."""
        v = np.ascontiguousarray(v, dtype=np.float64)
        z = np.empty_like(v)
        self._L.orc_mg_apply(self._h, _p64(v), _p64(z))
        return z

    def pcg(self, rtol=1e-10, max_iters=100):
        """CG preconditioned with one V-cycle; returns (x, iters, relres)."""
        rel = C.c_double(0)
        it = self._L.orc_mg_pcg(self._h, rtol, max_iters, C.byref(rel))
        return self.get_vec(0, "u"), int(it), rel.value

    def time_vcycles(self, n):
        return self._L.orc_mg_time_vcycles(self._h, n)
