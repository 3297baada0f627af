"""V-cycle on the strength-based C/F hierarchy (amg_hip_create_rs) on the device: general CSR
transfer kernels, SELL / dictionary level matrices of irregular rows, wide banded coarse solve.
NO counterpart in the reference ("parity unpinned" against it by construction): pinned to the
oracle twin -- same hierarchy bit for bit (tests/test_rs_coarsening.py), same V-cycle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def csc(A):
    return A.colptr, A.rowind, A.val


@pytest.mark.parametrize("sm", ["spgs", "jacobi", "multicolor"])
@pytest.mark.parametrize("n,dim", [(48, 2), (10, 3)])
def test_rs_vcycle_matches_oracle_twin(amg, oracle, sm, n, dim):
    A, b = oracle.laplacian(n, dim), oracle.rhs(n, dim)
    Ps = oracle.ruge_stueben_hierarchy(A, 12, 0.25, 40)
    kw_o = {"spgs": dict(smoother=oracle.SM_SPGS, smoother_iters=1),
            "jacobi": dict(smoother=oracle.SM_TRUE_JACOBI, smoother_iters=2, omega=0.6),
            "multicolor": dict(smoother=oracle.SM_MULTICOLOR, smoother_iters=1)}[sm]
    kw_p = {"spgs": dict(smoother=amg.SM_SPGS, smoother_iters=1),
            "jacobi": dict(smoother=amg.SM_JACOBI, smoother_iters=2, omega=0.6),
            "multicolor": dict(smoother=amg.SM_MULTICOLOR_GS, smoother_iters=1)}[sm]
    ref = oracle.Multigrid(A, b, len(Ps) + 1, transfers=Ps, **kw_o)
    mg = amg.Multigrid.ruge_stueben(*csc(A), b, 12, 0.25, 40, exact_coarse_solve=True, exact_gs=True, **kw_p)
    L = mg.n_levels
    assert L == len(Ps) + 1
    if sm == "multicolor":
        for l in range(L):
            col, nc = mg.get_colors(l)
            ref.set_colors(l, col, nc)
    r0 = ref.rss()
    for c in range(4):
        ref.vcycle()
        mg.vcycle()
        for l in range(L - 1):
            assert np.array_equal(mg.get_soln(l), ref.get_vec(l, "u")), (c, l)
            assert np.array_equal(mg.get_rhs(l), ref.get_vec(l, "f")), (c, l)
        assert abs(mg.rss() - ref.rss()) <= 1e-11 * ref.rss()
    assert ref.rss() < (1e-3 if sm != "jacobi" else 0.2) ** 2 * r0     # it is a fast iteration
    mg.close()


def test_rs_at_size_converges_and_beats_the_reference_coarsening(amg):
    """512^2: residual reduction per V-cycle with the reference's default smoother; the
    reference's own 6-level flat-index hierarchy needs ~10x the cycles for the same reduction."""
    n = 512
    cp, ri, v = amg.laplacian(n)
    b = amg.rhs(n)
    mg = amg.Multigrid.ruge_stueben(cp, ri, v, b, 20, 0.25, 500, smoother=amg.SM_MULTICOLOR_GS)
    assert 5 <= mg.n_levels <= 14 and mg.get_n_dofs(mg.n_levels - 1) <= 500
    assert mg.get_n_dofs(1) == n * n // 2
    r = [mg.rss()]
    for _ in range(8):
        mg.vcycle()
        r.append(mg.rss())
    fac = [(r[i + 1] / r[i]) ** 0.5 for i in range(8)]
    assert max(fac[1:]) < 0.25, fac
    mg.close()
    ref = amg.Multigrid(cp, ri, v, b, 6, smoother=amg.SM_MULTICOLOR_GS)
    r0 = ref.rss()
    ref.vcycle(8)
    assert (ref.rss() / r0) ** (0.5 / 8) > 2 * max(fac[1:])
    ref.close()


@pytest.mark.parametrize("n,dim", [(35, 2), (13, 3)])
def test_multicolor_sell_path_small_colours(amg, oracle, n, dim):
    """Regression: the SELL form of the colour sweep ran its last 256-thread workgroup on into
    the storage rows of the NEXT colour whenever a colour's padded row count was not a multiple
    of 256 (found with the irregular RS levels above).  Reference coarsening, SELL layout."""
    A, b = oracle.laplacian(n, dim), oracle.rhs(n, dim)
    L = 3
    ref = oracle.Multigrid(A, b, L, smoother=oracle.SM_MULTICOLOR, smoother_iters=1)
    mg = amg.Multigrid(*csc(A), b, L, smoother=amg.SM_MULTICOLOR_GS, smoother_iters=1,
                       layout=amg.LAYOUT_SELL, exact_coarse_solve=True)
    for l in range(L):
        col, nc = mg.get_colors(l)
        ref.set_colors(l, col, nc)
    for c in range(3):
        ref.vcycle()
        mg.vcycle()
        for l in range(L - 1):
            assert np.array_equal(mg.get_soln(l), ref.get_vec(l, "u")), (c, l)
    mg.close()
