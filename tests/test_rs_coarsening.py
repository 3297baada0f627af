"""Strength-based C/F coarsening (amg_hip_create_rs).  NO counterpart in the
reference (its README.md:104-109 names Ruge-Stueben as the road not taken): "parity unpinned"
against the reference by construction; pinned here to the oracle twin (oracle.ruge_stueben_P, a
plain-Python restatement of the textbook method) and to properties of the method itself.
Host-only solver objects: no device needed."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "algebraic-multigrid_amd"))


def csc(A):
    return A.colptr, A.rowind, A.val


@pytest.mark.parametrize("n,dim,theta", [(32, 2, 0.25), (37, 2, 0.25), (24, 2, 0.5), (9, 3, 0.25)])
def test_hierarchy_equals_oracle_twin_bit_for_bit(oracle, n, dim, theta):
    import amg_ctypes as amg
    A, b = oracle.laplacian(n, dim), oracle.rhs(n, dim)
    Ps = oracle.ruge_stueben_hierarchy(A, 12, theta, 30)
    mg = amg.Multigrid.ruge_stueben(*csc(A), b, 12, theta, 30, host_only=True)
    assert mg.n_levels == len(Ps) + 1 >= 3
    ref = oracle.Multigrid(A, b, len(Ps) + 1, transfers=Ps)
    for l, P in enumerate(Ps):
        assert mg.get_n_dofs(l + 1) == P.cols < mg.get_n_dofs(l)
        cp, ri, v = mg.get_transfer(l, "P")
        assert np.array_equal(cp, P.colptr) and np.array_equal(ri, P.rowind)     # structure: exact
        assert np.array_equal(v, P.val)                                          # weights: same bits
        R = P.transpose()
        cp, ri, v = mg.get_transfer(l, "R")
        assert np.array_equal(cp, R.colptr) and np.array_equal(ri, R.rowind) and np.array_equal(v, R.val)
    for l in range(mg.n_levels):                # Galerkin chain R (A P), Eigen's order
        M = ref.level_matrix(l)
        cp, ri, v = mg.get_coefficient_matrix(l)
        assert np.array_equal(cp, M.colptr) and np.array_equal(ri, M.rowind) and np.array_equal(v, M.val)
    mg.close()


def test_splitting_and_interpolation_properties(oracle):
    """5-point Poisson: the first pass yields the red-black splitting on level 0; every F point
    interpolates from strong C neighbours with positive weights; rows of interior points (zero
    row sum in A) sum to exactly 1 -- constants are reproduced."""
    n = 32
    A = oracle.laplacian(n)
    P, is_c = oracle.ruge_stueben_P(A, 0.25)
    ij = np.arange(n * n)
    red = ((ij // n) + (ij % n)) % 2 == 0
    assert np.array_equal(is_c, red) or np.array_equal(is_c, ~red)
    assert P.cols == n * n // 2
    Pr = P.transpose()                          # rows of P
    S = A.to_scipy().tocsr()
    rowsum_A = np.asarray(S.sum(axis=1)).ravel()
    for i in range(n * n):
        w = Pr.val[Pr.colptr[i]:Pr.colptr[i + 1]]
        if is_c[i]:
            assert w.tolist() == [1.0]
        else:
            assert w.size >= 2 and (w > 0).all()
            if abs(rowsum_A[i]) < 1e-9:
                assert abs(w.sum() - 1.0) < 1e-15
            else:
                assert w.sum() < 1.0            # next to the Dirichlet boundary


def test_vcycle_on_the_strength_based_hierarchy_converges_fast(oracle):
    """The point of the method: the reference's flat-index coarsening contracts by 0.36 (3
    levels) to 0.73 (6 levels) per cycle on this problem; the C/F hierarchy by < 0.15."""
    n = 64
    A, b = oracle.laplacian(n), oracle.rhs(n)
    Ps = oracle.ruge_stueben_hierarchy(A, 12, 0.25, 40)
    ref = oracle.Multigrid(A, b, len(Ps) + 1, transfers=Ps)     # SparseGaussSeidel(), 1 iteration
    r0 = ref.rss()
    fac = []
    for _ in range(6):
        ref.vcycle()
        r = ref.rss()
        fac.append((r / r0) ** 0.5)
        r0 = r
    assert max(fac[1:]) < 0.15, fac
    x, _ = oracle.band_solve(A, b)
    assert np.linalg.norm(ref.get_vec(0, "u") - x) <= 1e-6 * np.linalg.norm(x)


def test_argument_checks_and_stopping_rules(oracle):
    import amg_ctypes as amg
    A, b = oracle.laplacian(16), oracle.rhs(16)
    with pytest.raises(ValueError):
        amg.Multigrid.ruge_stueben(*csc(A), b, 5, theta=1.5, host_only=True)
    with pytest.raises(ValueError):
        amg.Multigrid.ruge_stueben(*csc(A), b, 5, min_coarse=0, host_only=True)
    mg = amg.Multigrid.ruge_stueben(*csc(A), b, 2, 0.25, 1, host_only=True)     # level cap
    assert mg.n_levels == 2
    mg.close()
    mg = amg.Multigrid.ruge_stueben(*csc(A), b, 10, 0.25, 1000, host_only=True)  # already small
    assert mg.n_levels == 1
    mg.close()
    # a diagonal matrix has no strong couplings: nothing to coarsen
    n = 10
    cp = np.arange(n + 1, dtype=np.int32)
    mg = amg.Multigrid.ruge_stueben(cp, np.arange(n, dtype=np.int32), -np.ones(n), np.ones(n), 5, 0.25, 1,
                                    host_only=True)
    assert mg.n_levels == 1
    mg.close()


def test_c_abi_argument_validation_without_device():
    import ctypes as C
    import amg_ctypes as amg
    lib = amg.lib()
    info = amg.SlabInfo()
    # slab plan: pure host arithmetic, every refusal is a status, never a crash
    assert lib.amg_hip_slab_plan(4096, 0, 0, 4, C.byref(info)) == amg.EINVAL        # world < 1
    assert lib.amg_hip_slab_plan(4096, 3, 2, 4, C.byref(info)) == amg.EINVAL        # rank >= world
    assert lib.amg_hip_slab_plan(4096, 0, 2, 0, C.byref(info)) == amg.EINVAL        # no slab level
    assert lib.amg_hip_slab_plan(4096, 0, 2, 9, C.byref(info)) == amg.EINVAL        # > AMG_HIP_SLAB_MAX_LEVELS
    assert lib.amg_hip_slab_plan(4096, 0, 2, 4, None) == amg.EINVAL
    assert lib.amg_hip_slab_plan(40, 0, 2, 4, C.byref(info)) == amg.EUNSUPPORTED     # 20 lines per rank < 23
    assert b"halo depth" in lib.amg_hip_last_error()
    assert lib.amg_hip_slab_plan(4096, 1, 2, 1, C.byref(info)) == amg.OK and info.halo_lines == 5
    assert lib.amg_hip_slab_run(None, 1) == amg.EINVAL
    assert lib.amg_hip_slab_setup(None, 0, 1, -1, C.byref(info)) == amg.EINVAL
    # create_rs: null arrays / null out handle
    h = C.c_void_p()
    o = amg.Options()
    lib.amg_hip_default_options(C.byref(o))
    o.host_only = 1
    assert lib.amg_hip_create_rs(4, None, None, None, None, 3, 0.25, 10, C.byref(o), C.byref(h)) == amg.EINVAL
    assert lib.amg_hip_create_rs(4, None, None, None, None, 3, 0.25, 10, C.byref(o), None) == amg.EINVAL


def test_sign_convention_and_nonsymmetric_input(oracle):
    """Couplings are measured against the sign of the diagonal: the reference's negative definite
    operator and its negation give the same splitting and the same interpolation weights.  A
    row-scaled (non-symmetric) matrix still yields the product == twin hierarchy."""
    import amg_ctypes as amg
    n = 20
    A, b = oracle.laplacian(n), oracle.rhs(n)
    P1, c1 = oracle.ruge_stueben_P(A, 0.25)
    An = oracle.CSC(A.rows, A.cols, A.colptr, A.rowind, -A.val)
    P2, c2 = oracle.ruge_stueben_P(An, 0.25)
    assert np.array_equal(c1, c2) and np.array_equal(P1.rowind, P2.rowind) and np.array_equal(P1.val, P2.val)
    # row scaling D A (D diagonal, positive): strength ratios per row unchanged -> same C/F split
    S = A.to_scipy().tocsr()
    d = 1.0 + 0.5 * np.arange(S.shape[0]) / S.shape[0]
    import scipy.sparse as sp
    M = (sp.diags(d) @ S).tocsc()
    M.sort_indices()
    As = oracle.CSC(A.rows, A.cols, M.indptr, M.indices, M.data)
    Ps = oracle.ruge_stueben_hierarchy(As, 6, 0.25, 30)
    P3, c3 = oracle.ruge_stueben_P(As, 0.25)
    assert np.array_equal(c3, c1)
    mg = amg.Multigrid.ruge_stueben(As.colptr, As.rowind, As.val, b, 6, 0.25, 30, host_only=True)
    assert mg.n_levels == len(Ps) + 1
    for l, P in enumerate(Ps):
        cp, ri, v = mg.get_transfer(l, "P")
        assert np.array_equal(cp, P.colptr) and np.array_equal(ri, P.rowind) and np.array_equal(v, P.val)
    mg.close()
