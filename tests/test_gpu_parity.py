"""GPU parity tests: every HIP kernel and the whole V-cycle, called through the C
ABI (libamg_hip.so), against the CPU oracle on the same inputs.

Bars (BASELINE.json north_star): level structure / C-F indices bit-exact;
residual norm and solution within 1e-10 relative.  Most kernels are in fact
bit-exact (same summation order, no FMA, IEEE divide) and are tested as such;
where a reduction order differs (rss tree sum) the tolerance is written out.
Nothing here reads /root/reference.
"""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(autouse=True, params=["csr", "sell", "sell_idx32", "dict", "dict_r1", "dict_words"])
def layout(request, amg):
    """Every test runs once per device layout / kernel variant: level matrices as plain
    CSR (LDS-staged K-CSR kernel), as SELL-64 panels with 16-bit relative column
    indices, as SELL-64 with plain int32 columns, and dictionary-coded (K-Dict, the
    default wherever a matrix qualifies, SELL otherwise) with two rows per lane, with
    one, and without the second-level row types.  Results must be bit-identical in all
    of them."""
    amg.set_default_layout(amg.LAYOUT_CSR if request.param == "csr" else
                           amg.LAYOUT_DICT if request.param.startswith("dict") else amg.LAYOUT_SELL)
    amg.set_dict_rows(1 if request.param == "dict_r1" else 2)
    amg.set_row_types(request.param != "dict_words")   # dict_words: first-level coding only
    amg.set_index16(request.param != "sell_idx32")
    yield request.param
    amg.set_default_layout(amg.LAYOUT_AUTO)
    amg.set_index16(True)
    amg.set_dict_rows(2)
    amg.set_row_types(True)


def csc(A):
    return A.colptr, A.rowind, A.val


def rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


@pytest.fixture(scope="module")
def mats(oracle):
    """A few level matrices of real hierarchies: 5-point fine, 9-point coarse
    with structural zeros (level 1) and flat-index-smeared coarse levels."""
    out = {}
    A, b = oracle.laplacian(67), oracle.rhs(67)     # odd n: ragged line ends
    mg = oracle.Multigrid(A, b, 5)
    for l in range(5):
        out[f"p67_l{l}"] = mg.level_matrix(l)
    out["p2"] = oracle.laplacian(2)
    out["p35"] = oracle.laplacian(35)
    out["p3d_9"] = oracle.laplacian(9, dim=3)
    return out


def test_residual_bit_exact(amg, oracle, mats):
    rng = np.random.default_rng(1)
    for name, A in mats.items():
        u, f = rng.standard_normal(A.rows), rng.standard_normal(A.rows)
        got = amg.residual(*csc(A), u, f)
        assert np.array_equal(got, oracle.residual(A, u, f)), name


def test_residual_nonsymmetric_uses_rows_of_A(amg, oracle):
    # r = f - A u must use A (not A^T) even when A != A^T
    import scipy.sparse as sp
    rng = np.random.default_rng(2)
    S = sp.random(300, 300, density=0.02, random_state=3, format="csc") + sp.eye(300, format="csc") * 4
    S.sort_indices()
    A = oracle.CSC(300, 300, S.indptr, S.indices, S.data)
    u, f = rng.standard_normal(300), rng.standard_normal(300)
    assert np.array_equal(amg.residual(*csc(A), u, f), oracle.residual(A, u, f))


def test_long_rows_take_the_chunked_path(amg, oracle):
    # rows far longer than one LDS chunk (arrowhead matrix): generic chunk loop
    import scipy.sparse as sp
    n = 6000
    rng = np.random.default_rng(5)
    D = sp.eye(n, format="lil") * 3.0
    D[0, :] = rng.standard_normal(n)
    D[:, 0] = rng.standard_normal((n, 1))
    D[n // 2, :] = rng.standard_normal(n)
    S = D.tocsc()
    S.sort_indices()
    A = oracle.CSC(n, n, S.indptr, S.indices, S.data)
    u, f = rng.standard_normal(n), rng.standard_normal(n)
    assert np.array_equal(amg.residual(*csc(A), u, f), oracle.residual(A, u, f))


def test_transfers_bit_exact(amg, oracle):
    rng = np.random.default_rng(3)
    for n_h in (7, 24, 1225, 612, 4096 * 3 + 1):
        n_H = oracle.n_H_from_n_h(n_h)
        P = oracle.make_P(n_h, n_H)
        R = P.transpose()
        r = rng.standard_normal(n_h)
        uH = rng.standard_normal(n_H)
        uh = rng.standard_normal(n_h)
        want_f = oracle.spmv(R, r)
        want_u = uh + oracle.spmv(P, uH)
        # matrix-free stencil kernels
        assert np.array_equal(amg.linear_restrict(n_h, n_H, r), want_f)
        assert np.array_equal(amg.linear_prolong_add(n_h, n_H, uH, uh), want_u)
        # generic CSR path (what a custom InterpolatorBase gets)
        assert np.array_equal(amg.spmv(R.rows, R.cols, *csc(R), r), want_f)
        assert np.array_equal(uh + amg.spmv(P.rows, P.cols, *csc(P), uH), want_u)
    # the reference's own smoke sizes (testlib.cpp:121,127): nH not from the formula
    for n_h, n_H in ((7, 3), (24, 11)):
        P = oracle.make_P(n_h, n_H)
        r = rng.standard_normal(n_h)
        assert np.array_equal(amg.linear_restrict(n_h, n_H, r), oracle.spmv(P.transpose(), r))


def test_rss_matches_sequential_sum(amg, oracle, mats):
    rng = np.random.default_rng(4)
    for name in ("p67_l0", "p67_l1", "p35"):
        A = mats[name]
        u, b = rng.standard_normal(A.rows), rng.standard_normal(A.rows)
        got, want = amg.rss(*csc(A), u, b), oracle.rss(A, u, b)
        # per-row terms are bit-identical; only the reduction tree differs from
        # the reference's sequential sum: O(eps*sqrt(N)) relative
        assert abs(got - want) <= 1e-13 * want, name


def test_coarse_solve_bit_exact(amg, oracle, mats):
    rng = np.random.default_rng(6)
    for name in ("p2", "p67_l4", "p67_l3", "p35"):
        A = mats[name]
        f = rng.standard_normal(A.rows)
        x, w = amg.coarse_solve(*csc(A), f)
        want, w_ref = oracle.band_solve(A, f)
        assert w == w_ref, name
        assert np.array_equal(x, want), name
        dense = np.linalg.solve(A.to_scipy().toarray(), f)
        assert rel(x, dense) < 1e-11, name


def test_spgs_sweeps_bit_exact(amg, oracle, mats):
    rng = np.random.default_rng(7)
    for name, A in mats.items():
        u, b = rng.standard_normal(A.rows), rng.standard_normal(A.rows)
        for d in (+1, -1):
            got = amg.spgs_sweep(d, *csc(A), u, b)
            assert np.array_equal(got, oracle.spgs_sweep(d, A, u, b)), (name, d)


def test_reference_smoothers_tiny_system(amg, oracle, mats):
    # testlib.cpp:74-107: Jacobi(100), SOR(100), SpGS(100) on the 4-dof system,
    # isApprox(exact, 1e-9); and bit-exact against the oracle
    A, b = mats["p2"], oracle.rhs(2)
    exact = np.linalg.solve(A.to_scipy().toarray(), b)
    for kind, okind in ((amg.SM_REF_JACOBI, oracle.SM_REF_JACOBI), (amg.SM_SOR, oracle.SM_SOR),
                        (amg.SM_SPGS, oracle.SM_SPGS)):
        u, it, conv = amg.smooth(kind, *csc(A), np.zeros(4), b, n_iters=100, tol=1e-9, every=100)
        want, it_ref, conv_ref = oracle.smooth(okind, A, np.zeros(4), b, n_iters=100, tol=1e-9, every=100)
        assert np.array_equal(u, want) and it == it_ref
        d2 = np.dot(u - exact, u - exact)
        assert d2 <= 1e-18 * min(np.dot(u, u), np.dot(exact, exact))


def test_sor_and_ref_jacobi_bit_exact(amg, oracle, mats):
    rng = np.random.default_rng(8)
    for name in ("p35", "p67_l1", "p67_l2"):
        A = mats[name]
        u0, b = rng.standard_normal(A.rows), rng.standard_normal(A.rows)
        for omega in (1.0, 1.3):
            u, _, _ = amg.smooth(amg.SM_SOR, *csc(A), u0, b, n_iters=3, omega=omega, every=100)
            want, _, _ = oracle.smooth(oracle.SM_SOR, A, u0, b, n_iters=3, omega=omega, every=100)
            assert np.array_equal(u, want), (name, omega)
        u, _, _ = amg.smooth(amg.SM_REF_JACOBI, *csc(A), u0, b, n_iters=3, every=100)
        want, _, _ = oracle.smooth(oracle.SM_REF_JACOBI, A, u0, b, n_iters=3, every=100)
        assert np.array_equal(u, want), name


def test_sor_omega_validation(amg, oracle, mats):
    # testlib.cpp:65-71
    A, b = mats["p2"], oracle.rhs(2)
    for bad in (-0.01, 2.01):
        with pytest.raises(ValueError, match=r"`omega` must be in \[0, 2\]"):
            amg.smooth(amg.SM_SOR, *csc(A), np.zeros(4), b, omega=bad, every=100)


def test_true_jacobi_bit_exact(amg, oracle, mats):
    rng = np.random.default_rng(9)
    for name, A in mats.items():
        u0, b = rng.standard_normal(A.rows), rng.standard_normal(A.rows)
        for omega, sweeps in ((0.8, 2), (1.0, 3)):
            u, _, _ = amg.smooth(amg.SM_JACOBI, *csc(A), u0, b, n_iters=sweeps, omega=omega)
            want, _, _ = oracle.smooth(oracle.SM_TRUE_JACOBI, A, u0, b, n_iters=sweeps, omega=omega)
            assert np.array_equal(u, want), (name, omega)


def test_spgs_standalone_kat3(amg, oracle):
    # testlib.cpp:188-196: SparseGaussSeidel(1e-9, 100, 1000) on 35^2
    A, b = oracle.laplacian(35), oracle.rhs(35)
    u, it, conv = amg.smooth(amg.SM_SPGS, *csc(A), np.zeros(1225), b, n_iters=1000, tol=1e-9, every=100)
    assert it == 900 and conv                         # "SPGS converged after 900 iterations."
    err = oracle.rss(A, u, b)
    assert f"{err:.5e}" == "8.69692e-10"             # "SPGS error: 8.69692e-10"
    want, _, _ = oracle.smooth(oracle.SM_SPGS, A, np.zeros(1225), b, n_iters=1000, tol=1e-9, every=100)
    assert np.array_equal(u, want)


def hierarchy_equal(amg_mg, ref, n_levels):
    for l in range(n_levels):
        assert amg_mg.get_n_dofs(l) == ref.n_dofs(l)
        cp, ri, v = amg_mg.get_coefficient_matrix(l)
        A = ref.level_matrix(l)
        assert np.array_equal(cp, A.colptr), l      # structure: bit-exact
        assert np.array_equal(ri, A.rowind), l
        assert np.array_equal(v.view(np.int64), A.val.view(np.int64)), l   # values: bit-exact
    for l in range(n_levels - 1):
        for w in ("P", "R"):
            cp, ri, v = amg_mg.get_transfer(l, w)
            T = ref.transfer(l, w)
            assert np.array_equal(cp, T.colptr) and np.array_equal(ri, T.rowind)
            assert np.array_equal(v, T.val)


def test_level_structure_bit_exact(amg, oracle):
    # KAT-1 + KAT-5: sizes, C/F (P/R patterns), Galerkin patterns incl. structural zeros
    for n, L in ((35, 8), (128, 3), (67, 5)):
        A, b = oracle.laplacian(n), oracle.rhs(n)
        ref = oracle.Multigrid(A, b, L)
        mg = amg.Multigrid(*csc(A), b, L)
        hierarchy_equal(mg, ref, L)
        if n == 35:
            assert [mg.get_n_dofs(l) for l in range(8)] == [1225, 612, 305, 152, 75, 37, 18, 8]
        if n == 128:
            got = [(mg.get_n_dofs(l), mg.get_coefficient_matrix(l)[1].size) for l in range(3)]
            assert got == [(16384, 81408), (8191, 73333), (4095, 36661)]
        mg.close()


def test_device_galerkin_matches_host_and_oracle(amg, oracle):
    """K-Galerkin (R (A P) on the device for the linear interpolation pair) against the
    host SpGEMM and the oracle: pattern incl. structural zeros, and values, bit for bit,
    on every level; 2-D (odd and even line lengths) and 3-D."""
    for n, dim, L in ((67, 2, 7), (64, 2, 6), (128, 2, 5), (13, 3, 6), (300, 2, 4)):
        A, b = oracle.laplacian(n, dim=dim), oracle.rhs(n, dim=dim)
        ref = oracle.Multigrid(A, b, L)
        dev = amg.Multigrid(*csc(A), b, L, smoother=amg.SM_JACOBI, omega=0.6)
        host = amg.Multigrid(*csc(A), b, L, smoother=amg.SM_JACOBI, omega=0.6, host_galerkin=True)
        for l in range(L):
            want = ref.level_matrix(l)
            for mg in (dev, host):
                cp, ri, va = mg.get_coefficient_matrix(l)
                assert np.array_equal(cp, want.colptr), (n, dim, l)
                assert np.array_equal(ri, want.rowind), (n, dim, l)
                assert np.array_equal(va, want.val), (n, dim, l)
        dev.close()
        host.close()


def test_vcycle_config1_bit_exact_and_golden(amg, oracle):
    """BASELINE config 1: 128^2, 3 levels, SparseGaussSeidel() (nu1=nu2=2 sweeps).
    GPU V-cycle == oracle V-cycle bit for bit, every level vector, 12 cycles;
    rss trajectory == committed golden fixture to 1e-10 relative."""
    n, L = 128, 3
    A, b = oracle.laplacian(n), oracle.rhs(n)
    ref = oracle.Multigrid(A, b, L)
    mg = amg.Multigrid(*csc(A), b, L, keep_residual=True)
    gold = json.load(open(os.path.join(GOLD, "kat_config1.json")))
    for c in range(12):
        ref.vcycle()
        mg.vcycle()
        for l in range(L):
            assert np.array_equal(mg.get_soln(l), ref.get_vec(l, "u")), (c, l)
            assert np.array_equal(mg.get_rhs(l), ref.get_vec(l, "f")), (c, l)
            assert np.array_equal(mg.get_residual(l), ref.get_vec(l, "r")), (c, l)
        got = mg.rss()
        assert abs(got - gold["rss"][c]) <= 1e-10 * gold["rss"][c], c
    u = mg.get_soln(0)
    assert abs(np.linalg.norm(u) - gold["u_norm2"]) <= 1e-10 * gold["u_norm2"]
    assert abs(u[8255] - gold["u_8255"]) <= 1e-10 * abs(gold["u_8255"])
    mg.close()


def test_solve_kat2_35_iterations(amg, oracle, capsys):
    # testlib.cpp:158-159,203-206,212
    A, b = oracle.laplacian(35), oracle.rhs(35)
    mg = amg.Multigrid(*csc(A), b, 8, tolerance=1e-9, compute_error_every_n_iters=5, n_iters=100)
    u, it, conv, last = mg.solve()
    assert "AMG converged after 35 iterations." in capsys.readouterr().out
    assert it == 35 and conv
    err = oracle.rss(A, u, b)
    assert f"{err:.5e}" == "7.19199e-11"            # "AMG error: 7.19199e-11"
    assert err < mg.get_tolerance()
    spgs_u, _, _ = oracle.smooth(oracle.SM_SPGS, A, np.zeros(1225), b, n_iters=1000, tol=1e-9, every=100)
    d2 = np.dot(u - spgs_u, u - spgs_u)
    assert d2 <= 1e-12 * min(np.dot(u, u), np.dot(spgs_u, spgs_u))   # isApprox(.., 1e-6)
    ref = oracle.Multigrid(A, b, 8)
    ref.solve(1e-9, 5, 100)
    assert np.array_equal(u, ref.get_vec(0, "u"))
    mg.close()


@pytest.mark.parametrize("graph", [True, False])
def test_vcycle_true_jacobi_bit_exact(amg, oracle, graph):
    n, L = 96, 5
    A, b = oracle.laplacian(n), oracle.rhs(n)
    ref = oracle.Multigrid(A, b, L, smoother=oracle.SM_TRUE_JACOBI, smoother_iters=2, omega=0.8)
    mg = amg.Multigrid(*csc(A), b, L, smoother=amg.SM_JACOBI, smoother_iters=2, omega=0.8,
                       use_graph=graph)
    r_prev = None
    for c in range(6):
        ref.vcycle()
        mg.vcycle()
        assert np.array_equal(mg.get_soln(0), ref.get_vec(0, "u")), c
        r = mg.rss()
        assert abs(r - ref.rss()) <= 1e-12 * ref.rss()
        if r_prev is not None:
            assert r < 0.5 * r_prev        # the build-side smoother does converge
        r_prev = r
    mg.close()


def test_vcycle_odd_jacobi_sweeps(amg, oracle):
    A, b = oracle.laplacian(40), oracle.rhs(40)
    ref = oracle.Multigrid(A, b, 4, smoother=oracle.SM_TRUE_JACOBI, smoother_iters=3, omega=0.7)
    mg = amg.Multigrid(*csc(A), b, 4, smoother=amg.SM_JACOBI, smoother_iters=3, omega=0.7)
    for _ in range(3):
        ref.vcycle()
        mg.vcycle()
    assert np.array_equal(mg.get_soln(0), ref.get_vec(0, "u"))
    mg.close()


def test_vcycle_sor_and_ref_jacobi_smoothers(amg, oracle):
    A, b = oracle.laplacian(35), oracle.rhs(35)
    for kind, okind, om in ((amg.SM_SOR, oracle.SM_SOR, 1.2), (amg.SM_REF_JACOBI, oracle.SM_REF_JACOBI, 1.0)):
        ref = oracle.Multigrid(A, b, 6, smoother=okind, smoother_iters=2, omega=om)
        mg = amg.Multigrid(*csc(A), b, 6, smoother=kind, smoother_iters=2, omega=om)
        for _ in range(4):
            ref.vcycle()
            mg.vcycle()
        assert np.array_equal(mg.get_soln(0), ref.get_vec(0, "u")), kind
        mg.close()


def test_custom_interpolator_path(amg, oracle):
    """A user InterpolatorBase: P/R handed over as CSC.  Passing exactly the
    linear operators must give the built-in result (and take the stencil path);
    a scaled R exercises the generic CSR transfer kernels."""
    n, L = 35, 5
    A, b = oracle.laplacian(n), oracle.rhs(n)
    ref = oracle.Multigrid(A, b, L)
    tr = [(csc(ref.transfer(l, "P")), csc(ref.transfer(l, "R"))) for l in range(L - 1)]
    mg = amg.Multigrid(*csc(A), b, L, transfers=tr)
    mg2 = amg.Multigrid(*csc(A), b, L, transfers=tr, stencil_transfers=False)
    hierarchy_equal(mg, ref, L)
    for _ in range(3):
        ref.vcycle()
        mg.vcycle()
        mg2.vcycle()
    assert np.array_equal(mg.get_soln(0), ref.get_vec(0, "u"))
    assert np.array_equal(mg2.get_soln(0), ref.get_vec(0, "u"))
    mg.close()
    mg2.close()
    # R = 0.5 * P^T (not the built-in): generic CSR transfer kernels.  Galerkin
    # A_H and f_H both scale by 0.5, so the coarse correction -- and the whole
    # iteration -- is the same up to rounding.
    tr2 = [((P[0], P[1], P[2]), (R[0], R[1], 0.5 * R[2])) for (P, R) in tr]
    mg3 = amg.Multigrid(*csc(A), b, L, transfers=tr2)
    mg3.vcycle(3)
    assert rel(mg3.get_soln(0), ref.get_vec(0, "u")) < 1e-12
    mg3.close()


def test_set_vec_and_persistence_of_level0_solution(amg, oracle):
    # level-0 u persists across cycles, coarse u is zeroed on the way down
    A, b = oracle.laplacian(35), oracle.rhs(35)
    ref = oracle.Multigrid(A, b, 4)
    mg = amg.Multigrid(*csc(A), b, 4)
    rng = np.random.default_rng(11)
    u0 = rng.standard_normal(1225)
    ref.set_vec(0, "u", u0)
    mg.set_vec(0, "u", u0)
    junk = rng.standard_normal(ref.n_dofs(1))
    ref.set_vec(1, "u", junk)
    mg.set_vec(1, "u", junk)
    ref.vcycle()
    mg.vcycle()
    assert np.array_equal(mg.get_soln(0), ref.get_vec(0, "u"))
    mg.close()


def test_config2_like_parity_1e10(amg, oracle):
    """BASELINE config 2 shape (6 levels, fp64, exact SpGS), at 256^2 so the CPU
    oracle finishes in seconds: solution and rss within 1e-10 relative (they are
    bit-exact), coarse half-bandwidth 9."""
    n, L = 256, 6
    A, b = oracle.laplacian(n), oracle.rhs(n)
    ref = oracle.Multigrid(A, b, L)
    mg = amg.Multigrid(*csc(A), b, L)
    assert mg.coarse_halfbw() == ref.coarse_halfbw()
    for c in range(4):
        ref.vcycle()
        mg.vcycle()
        assert rel(mg.get_soln(0), ref.get_vec(0, "u")) <= 1e-10
        assert abs(mg.rss() - ref.rss()) <= 1e-10 * ref.rss()
    assert np.array_equal(mg.get_soln(0), ref.get_vec(0, "u"))
    mg.close()


def test_3d_7point_vcycle(amg, oracle):
    # BASELINE config 5 shape at 16^3: build-side generator, oracle twin only
    n, L = 16, 5
    A, b = oracle.laplacian(n, dim=3), oracle.rhs(n, dim=3)
    ref = oracle.Multigrid(A, b, L, smoother=oracle.SM_TRUE_JACOBI, smoother_iters=2, omega=0.8)
    mg = amg.Multigrid(*csc(A), b, L, smoother=amg.SM_JACOBI, smoother_iters=2, omega=0.8)
    for _ in range(3):
        ref.vcycle()
        mg.vcycle()
    assert np.array_equal(mg.get_soln(0), ref.get_vec(0, "u"))
    mg.close()


def test_multicolor_gs_vcycle_bit_exact(amg, oracle):
    """Build-side symmetric multicolour Gauss-Seidel (BASELINE config 4 smoother).
    Colouring comes from the product (greedy, row order); the oracle twin replays
    the same colours.  Level 0 of the 5-point operator must come out red-black."""
    n, L = 64, 5
    A, b = oracle.laplacian(n), oracle.rhs(n)
    mg = amg.Multigrid(*csc(A), b, L, smoother=amg.SM_MULTICOLOR_GS, smoother_iters=1)
    ref = oracle.Multigrid(A, b, L, smoother=oracle.SM_MULTICOLOR, smoother_iters=1)
    for l in range(L):
        color, nc = mg.get_colors(l)
        ref.set_colors(l, color, nc)
        # proper colouring of the numerically non-zero pattern
        S = ref.level_matrix(l).to_scipy().tocoo()
        m = (S.row != S.col) & (S.data != 0)
        assert not np.any(color[S.row[m]] == color[S.col[m]]), l
        assert color.min() == 0 and color.max() == nc - 1
        if l == 0:
            i, j = np.divmod(np.arange(n * n), n)
            assert nc == 2 and np.array_equal(color, (i + j) % 2)     # red-black
    rss = []
    for c in range(6):
        ref.vcycle()
        mg.vcycle()
        assert np.array_equal(mg.get_soln(0), ref.get_vec(0, "u")), c
        rss.append(mg.rss())
    assert rss[-1] < 1e-2 * rss[0]          # converges (oracle twin: 1.4e3 -> 4.8 in 6 cycles)
    mg.close()


def test_multicolor_gs_two_iterations_3d(amg, oracle):
    n, L = 12, 3
    A, b = oracle.laplacian(n, dim=3), oracle.rhs(n, dim=3)
    mg = amg.Multigrid(*csc(A), b, L, smoother=amg.SM_MULTICOLOR_GS, smoother_iters=2)
    ref = oracle.Multigrid(A, b, L, smoother=oracle.SM_MULTICOLOR, smoother_iters=2)
    for l in range(L):
        color, nc = mg.get_colors(l)
        ref.set_colors(l, color, nc)
    for _ in range(3):
        ref.vcycle()
        mg.vcycle()
    assert np.array_equal(mg.get_soln(0), ref.get_vec(0, "u"))
    mg.close()


def test_jacobi_fusions_and_zero_pruning_are_bit_neutral(amg, oracle):
    """The true-Jacobi V-cycle shortcuts (first coarse pre-sweep from the diagonal
    alone, prolongation applied inside the first post-sweep, exact-zero entries
    dropped from the device copies) must not change a single bit."""
    n, L = 80, 6
    A, b = oracle.laplacian(n), oracle.rhs(n)
    ref = oracle.Multigrid(A, b, L, smoother=oracle.SM_TRUE_JACOBI, smoother_iters=2, omega=0.6)
    variants = [dict(), dict(no_fusion=True), dict(keep_structural_zeros=True), dict(fuse_prolong=True),
                dict(no_fusion=True, keep_structural_zeros=True), dict(stencil_transfers=False)]
    mgs = [amg.Multigrid(*csc(A), b, L, smoother=amg.SM_JACOBI, smoother_iters=2, omega=0.6,
                         keep_residual=True, **kw)
           for kw in variants]
    for c in range(4):
        ref.vcycle()
        for mg in mgs:
            mg.vcycle()
        for l in range(L):
            want = ref.get_vec(l, "u")
            for k, mg in enumerate(mgs):
                assert np.array_equal(mg.get_soln(l), want), (c, l, variants[k])
                assert np.array_equal(mg.get_residual(l), ref.get_vec(l, "r")), (c, l, variants[k])
    for mg in mgs:
        mg.close()
    # one sweep per smooth(): the fused sweep is also the last one
    ref = oracle.Multigrid(A, b, L, smoother=oracle.SM_TRUE_JACOBI, smoother_iters=1, omega=0.5)
    mg = amg.Multigrid(*csc(A), b, L, smoother=amg.SM_JACOBI, smoother_iters=1, omega=0.5)
    for _ in range(3):
        ref.vcycle()
        mg.vcycle()
    assert np.array_equal(mg.get_soln(0), ref.get_vec(0, "u"))
    mg.close()


def test_vcycle_nonsymmetric_operator(amg, oracle):
    """A != A^T: the residual uses the rows of A, the smoothers walk its CSC columns as
    rows (smoother.hpp:101-117), so the fused residual+restriction kernel and the fused
    sweep+prolongation kernel run on two different device copies.  Convection-diffusion-like
    band matrix, true Jacobi and the exact SpGS, every level vector against the oracle."""
    n = 5000
    cp = np.zeros(n + 1, dtype=np.int32)
    ri, va = [], []
    for j in range(n):                      # column j: rows j-70, j-1, j, j+1, j+70
        for i, v in ((j - 70, -1.0), (j - 1, -1.5), (j, 5.0), (j + 1, -0.5), (j + 70, -1.0)):
            if 0 <= i < n:
                ri.append(i)
                va.append(v + (0.125 if i == j and j % 3 == 0 else 0.0))
        cp[j + 1] = len(ri)
    A = oracle.CSC(n, n, cp, np.array(ri, dtype=np.int32), np.array(va))
    b = np.sin(0.01 * np.arange(n)) + 2.0
    L = 4
    for okind, kind, kw in ((oracle.SM_TRUE_JACOBI, amg.SM_JACOBI, dict(smoother_iters=2, omega=0.6)),
                            (oracle.SM_SPGS, amg.SM_SPGS, dict(smoother_iters=1))):
        ref = oracle.Multigrid(A, b, L, smoother=okind, **kw)
        mg = amg.Multigrid(*csc(A), b, L, smoother=kind, keep_residual=True, **kw)
        for c in range(3):
            ref.vcycle()
            mg.vcycle()
            for l in range(L):
                assert np.array_equal(mg.get_soln(l), ref.get_vec(l, "u")), (kind, c, l)
                assert np.array_equal(mg.get_rhs(l), ref.get_vec(l, "f")), (kind, c, l)
                assert np.array_equal(mg.get_residual(l), ref.get_vec(l, "r")), (kind, c, l)
        mg.close()


def test_residual_is_private_workspace_by_default(amg, oracle):
    """multigrid.hpp:107: level_to_residual has no getter.  Without keep_residual the fused
    residual+restriction kernel does not store r and the coarsest level skips its (dead)
    smoothing and residual; u and f are unaffected, reading r there is an error."""
    n, L = 80, 4
    A, b = oracle.laplacian(n), oracle.rhs(n)
    ref = oracle.Multigrid(A, b, L, smoother=oracle.SM_TRUE_JACOBI, smoother_iters=2, omega=0.6)
    mg = amg.Multigrid(*csc(A), b, L, smoother=amg.SM_JACOBI, smoother_iters=2, omega=0.6)
    for _ in range(3):
        ref.vcycle()
        mg.vcycle()
    for l in range(L):
        assert np.array_equal(mg.get_soln(l), ref.get_vec(l, "u"))
        assert np.array_equal(mg.get_rhs(l), ref.get_vec(l, "f"))
    with pytest.raises(amg.AmgHipError):   # coarsest: smoothing + residual are dead work there
        mg.get_residual(L - 1)
    lay, _ = mg.level_layout(0)
    if lay == amg.LAYOUT_DICT:        # the fusion needs the dictionary-coded layout
        with pytest.raises(amg.AmgHipError):
            mg.get_residual(0)
    mg.close()


def test_fused_level_kernels_at_tile_boundaries(amg, oracle):
    """The fused residual+restriction+first-coarse-sweep and sweep+prolongation kernels
    work on overlapping tiles of 256 or 512 rows (stride 254 / 510, one or two rows per
    lane from 4096 rows on): sizes around every boundary, odd and even, against the
    oracle bit for bit on every level vector (u, f, r)."""
    for n in (253, 254, 255, 256, 257, 509, 510, 511, 512, 513, 765, 1019, 1020, 1021, 1022,
              4095, 4096, 4097, 4605, 4606, 4607, 5117, 8191):
        cp = np.zeros(n + 1, dtype=np.int32)
        ri, va = [], []
        for j in range(n):        # tridiagonal, diagonally dominant, row-dependent values
            for i in (j - 1, j, j + 1):
                if 0 <= i < n:
                    ri.append(i)
                    va.append(2.5 + 0.25 * ((j * 7) % 5) if i == j else -1.0 - 0.125 * ((i + j) % 3))
            cp[j + 1] = len(ri)
        A = oracle.CSC(n, n, cp, np.array(ri, dtype=np.int32), np.array(va))
        b = np.cos(0.37 * np.arange(n)) + 1.5
        L = 3
        ref = oracle.Multigrid(A, b, L, smoother=oracle.SM_TRUE_JACOBI, smoother_iters=2, omega=0.6)
        mg = amg.Multigrid(*csc(A), b, L, smoother=amg.SM_JACOBI, smoother_iters=2, omega=0.6,
                           keep_residual=True)
        for c in range(2):
            ref.vcycle()
            mg.vcycle()
            for l in range(L):
                assert np.array_equal(mg.get_soln(l), ref.get_vec(l, "u")), (n, c, l)
                assert np.array_equal(mg.get_rhs(l), ref.get_vec(l, "f")), (n, c, l)
                assert np.array_equal(mg.get_residual(l), ref.get_vec(l, "r")), (n, c, l)
        mg.close()


def test_vcycle_random_band_operators(amg, oracle):
    """Seeded sweep over sizes, band offsets, sweep counts and level counts: symmetric band
    operators with a second band at +-k (the flat-index picture of a 2-D stencil with line
    length k), true Jacobi, every level's u and f against the oracle bit for bit."""
    rng = np.random.default_rng(20251004)
    for case in range(14):
        n = int(rng.integers(300, 20000))
        k = int(rng.integers(3, 200))
        L = int(rng.integers(2, 6))
        sweeps = int(rng.integers(1, 4))
        while (n >> (L - 1)) < 8:
            L -= 1
        cp = np.zeros(n + 1, dtype=np.int32)
        ri, va = [], []
        for j in range(n):
            for i, v in ((j - k, -1.0), (j - 1, -1.0 if j % k else 0.0), (j, 4.0 + 0.5 * (j % 2)),
                         (j + 1, -1.0 if (j + 1) % k else 0.0), (j + k, -1.0)):
                if 0 <= i < n and v != 0.0:
                    ri.append(i)
                    va.append(v)
            cp[j + 1] = len(ri)
        A = oracle.CSC(n, n, cp, np.array(ri, dtype=np.int32), np.array(va))
        b = 1.0 + rng.random(n)
        ref = oracle.Multigrid(A, b, L, smoother=oracle.SM_TRUE_JACOBI, smoother_iters=sweeps, omega=0.55)
        mg = amg.Multigrid(*csc(A), b, L, smoother=amg.SM_JACOBI, smoother_iters=sweeps, omega=0.55,
                           exact_coarse_solve=True)   # bit-exact bar: sequential substitution
        for c in range(2):
            ref.vcycle()
            mg.vcycle()
        for l in range(L):
            assert np.array_equal(mg.get_soln(l), ref.get_vec(l, "u")), (case, n, k, L, sweeps, l)
            assert np.array_equal(mg.get_rhs(l), ref.get_vec(l, "f")), (case, n, k, L, sweeps, l)
        mg.close()


def test_partitioned_coarse_solve(amg, oracle, mats):
    """opt.fast_coarse_solve: the banded LDL^T solve cut into partitions that are solved
    in parallel and coupled by a short boundary recurrence.  Same direct solve as the
    sequential substitution (and the oracle's), different rounding order: 1e-11 bar."""
    rng = np.random.default_rng(12)
    cases = [mats["p2"], mats["p35"], mats["p67_l3"], mats["p67_l4"]]
    big = oracle.Multigrid(oracle.laplacian(256), oracle.rhs(256), 5)
    cases += [big.level_matrix(3), big.level_matrix(4)]         # 8191 (w 33) and 4095 (w 17) dofs
    deep = oracle.Multigrid(oracle.laplacian(128), oracle.rhs(128), 11)
    cases.append(deep.level_matrix(10))                           # half-bandwidth 2, a few rows
    for A in cases:
        f = rng.standard_normal(A.rows)
        x, w, c = amg.coarse_solve_fast(*csc(A), f)
        want, w_ref = oracle.band_solve(A, f)
        assert w == w_ref and c >= max(w, 1)
        assert rel(x, want) < 1e-11, (A.rows, w, c, rel(x, want))
        S = A.to_scipy()
        assert np.linalg.norm(S @ x - f) <= 1e-10 * np.linalg.norm(f) * (1 + abs(S).max() * np.abs(x).max() / np.abs(f).max())


def test_vcycle_with_partitioned_coarse_solve_within_1e10(amg, oracle):
    # north-star tolerance (1e-10 relative) for solution and rss when the coarse solve is
    # the parallel one; everything else of the cycle stays bit-identical
    for n, L in ((128, 3), (200, 6)):
        A, b = oracle.laplacian(n), oracle.rhs(n)
        ref = oracle.Multigrid(A, b, L, smoother=oracle.SM_TRUE_JACOBI, smoother_iters=2, omega=0.6)
        mg = amg.Multigrid(*csc(A), b, L, smoother=amg.SM_JACOBI, smoother_iters=2, omega=0.6,
                           fast_coarse_solve=True)
        for c in range(6):
            ref.vcycle()
            mg.vcycle()
            assert rel(mg.get_soln(0), ref.get_vec(0, "u")) <= 1e-10, (n, c)
            assert abs(mg.rss() - ref.rss()) <= 1e-10 * ref.rss(), (n, c)
        mg.close()
