"""Full-size GPU checks (BASELINE.json sizes).  The oracle is too slow to replay
4096^2 in a test, so these use (a) the oracle where it still finishes in
seconds (config 2: 1024^2, 6 levels) and (b) size-independent properties:
bit-equality of the two device layouts, linearity of the residual, zero in ->
zero out, exactness of the transfers on linear data, monotone rss."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def csc(A):
    return A.colptr, A.rowind, A.val


def test_config2_full_size_vs_oracle(amg, oracle):
    """BASELINE config 2: 1024x1024, 6-level V-cycle, fp64, SparseGaussSeidel().
    GPU == oracle: level structure bit-exact, solution/rss within 1e-10 (bit-exact)."""
    n, L = 1024, 6
    A, b = oracle.laplacian(n), oracle.rhs(n)
    ref = oracle.Multigrid(A, b, L)
    # parity mode: exact lexicographic GS kernel and sequential coarse substitution (the
    # defaults at this size are the line-scan sweeps and the partitioned solve, 1e-10 class)
    mg = amg.Multigrid(*csc(A), b, L, exact_coarse_solve=True, exact_gs=True)
    sizes = [mg.get_n_dofs(l) for l in range(L)]
    assert sizes == [1048576, 524287, 262143, 131071, 65535, 32767]
    nnz = [mg.get_coefficient_matrix(l)[1].size for l in range(L)]
    assert nnz == [5238784, 4715509, 2357749, 1178869, 589429, 294709]        # KAT-5
    for l in (1, 5):
        cp, ri, v = mg.get_coefficient_matrix(l)
        R = ref.level_matrix(l)
        assert np.array_equal(cp, R.colptr) and np.array_equal(ri, R.rowind) and np.array_equal(v, R.val)
    assert mg.coarse_halfbw() == ref.coarse_halfbw() == 33
    for c in range(2):
        ref.vcycle()
        mg.vcycle()
        u, ur = mg.get_soln(0), ref.get_vec(0, "u")
        assert np.linalg.norm(u - ur) <= 1e-10 * np.linalg.norm(ur)
        assert abs(mg.rss() - ref.rss()) <= 1e-10 * ref.rss()
    assert np.array_equal(mg.get_soln(0), ref.get_vec(0, "u"))
    mg.close()


@pytest.fixture(scope="module")
def big(amg):
    n = 4096
    colptr, rowind, val = amg.laplacian(n)
    return n, colptr, rowind, val, amg.rhs(n)


def test_4096_layouts_agree_bitwise_and_rss_decreases(amg, big):
    """BASELINE config 3 (4096^2, true Jacobi): the dictionary-coded, the SELL-64 and
    the LDS-staged CSR kernels give the same bits over whole V-cycles; rss decreases monotonically
    after the first cycle; level sizes follow n_H = (n_h+1)/2 - 1."""
    n, cp, ri, v, b = big
    L = 16
    out = {}
    for lay in (amg.LAYOUT_SELL, amg.LAYOUT_CSR, amg.LAYOUT_DICT):
        mg = amg.Multigrid(cp, ri, v, b, L, smoother=amg.SM_JACOBI, smoother_iters=2, omega=0.6, layout=lay)
        if lay == amg.LAYOUT_SELL:
            sizes = [mg.get_n_dofs(l) for l in range(L)]
            want = [n * n]
            for _ in range(L - 1):
                want.append((want[-1] + 1) // 2 - 1)
            assert sizes == want and sizes[-1] == 511
        rss = []
        for _ in range(4):
            mg.vcycle()
            rss.append(mg.rss())
        assert all(rss[i + 1] < rss[i] for i in range(len(rss) - 1)), rss
        out[lay] = (mg.get_soln(0), rss)
        mg.close()
    for lay in (amg.LAYOUT_CSR, amg.LAYOUT_DICT):
        assert np.array_equal(out[amg.LAYOUT_SELL][0], out[lay][0])
        assert out[amg.LAYOUT_SELL][1] == out[lay][1]


def test_4096_residual_is_affine_and_exact_on_quadratics(amg, big):
    n, cp, ri, v, b = big
    N = n * n
    rng = np.random.default_rng(0)
    u1, u2 = rng.standard_normal(N), rng.standard_normal(N)
    z = np.zeros(N)
    r1 = amg.residual(cp, ri, v, u1, z)      # -A u1
    r2 = amg.residual(cp, ri, v, u2, z)
    r12 = amg.residual(cp, ri, v, u1 + u2, z)
    assert np.linalg.norm(r12 - (r1 + r2)) <= 1e-13 * np.linalg.norm(r12)
    assert np.array_equal(amg.residual(cp, ri, v, z, b), b)          # u = 0 -> r = f exactly
    # the 5-point operator is exact on u(x,y) = x^2 + y^2 in the interior: A u = 4
    h = 2.0 / (n + 1)
    x = -1.0 + h * np.arange(1, n + 1)
    U = (x[None, :] ** 2 + x[:, None] ** 2).ravel()
    Au = -amg.residual(cp, ri, v, U, z).reshape(n, n)
    assert np.abs(Au[1:-1, 1:-1] - 4.0).max() < 1e-5                # cancellation ~ eps / h^2


def test_4096_transfers_properties(amg, big):
    n = big[0]
    n_h = n * n
    n_H = (n_h + 1) // 2 - 1
    # restriction of a constant = 2*c (row sums 0.5+1+0.5); prolongation of a linear
    # coarse function is linear on the fine dofs it reaches
    r = np.full(n_h, 3.0)
    assert np.array_equal(amg.linear_restrict(n_h, n_H, r), np.full(n_H, 6.0))
    uH = np.arange(n_H, dtype=np.float64)
    got = amg.linear_prolong_add(n_h, n_H, uH, np.zeros(n_h))
    i = np.arange(1, n_h - 2)
    assert np.array_equal(got[i], (i - 1) / 2.0)
    assert got[n_h - 1] == 0.0                                       # last fine row never corrected (n_h even)


def test_4096_zero_rhs_stays_zero_and_rss_matches_norm(amg, big):
    n, cp, ri, v, b = big
    mg = amg.Multigrid(cp, ri, v, np.zeros(n * n), 16, smoother=amg.SM_JACOBI, smoother_iters=2, omega=0.6)
    mg.vcycle(2)
    assert not mg.get_soln(0).any()
    assert mg.rss() == 0.0
    mg.close()
    mg = amg.Multigrid(cp, ri, v, b, 16, smoother=amg.SM_JACOBI, smoother_iters=2, omega=0.6)
    got = mg.rss()                                                   # u = 0: rss = sum b^2
    assert abs(got - float(np.dot(b, b))) <= 1e-12 * got
    mg.close()


def test_4096_bench_object_equals_host_constructor_sell_cycle(amg, big):
    """VERDICT r2 #3a: the exact object bench.py times -- Multigrid.poisson(4096, 16): setup on the
    device end to end, dictionary rows, K-Patch legs on levels 0-3, fused pair kernels, K-BandChain
    -- against the host-array constructor with plain CSR panels (SELL-64, no K-Patch, no pair
    kernels): every level-0 bit equal after each of 3 cycles, and the same rss."""
    n, cp, ri, v, b = big
    L = 16
    dev = amg.Multigrid.poisson(n, L, smoother=amg.SM_JACOBI, smoother_iters=2, omega=0.6)
    ref = amg.Multigrid(cp, ri, v, b, L, smoother=amg.SM_JACOBI, smoother_iters=2, omega=0.6,
                        layout=amg.LAYOUT_SELL)
    assert dev.fine_sweep_info()[0].startswith("patch_down_kernel")        # K-Patch really is on
    assert ref.fine_sweep_info()[0].startswith("sell_kernel")
    assert [dev.get_n_dofs(l) for l in range(L)] == [ref.get_n_dofs(l) for l in range(L)]
    for c in range(3):
        dev.vcycle()
        ref.vcycle()
        assert np.array_equal(dev.get_soln(0), ref.get_soln(0)), c
        assert dev.rss() == ref.rss()
    for l in (1, 4, 9, 15):                                                 # K-Patch / K-Dict / pair / coarsest
        assert np.array_equal(dev.get_soln(l), ref.get_soln(l)), l
    dev.close()
    ref.close()


def test_config2_solve_with_default_smoother_matches_oracle(amg, oracle, capsys):
    """VERDICT r2 #3b: Multigrid::solve() (multigrid.hpp:311-337) at BASELINE config 2 with the
    reference's default smoother SparseGaussSeidel() -- on the device the K-GS-scan sweeps (the
    default from 65536 fine rows) and the partitioned coarse solve -- against the oracle's solve():
    same iteration count, same convergence verdict, final rss and solution within 1e-10."""
    n, L = 1024, 6
    A, b = oracle.laplacian(n), oracle.rhs(n)
    # the reference's 35^2 test converges in 35 cycles; this deep-grid instance contracts slowly
    # (DESIGN.md "Smoothers and convergence"), so the loop is cut at 40 cycles, checked every 10
    tol, every, iters = 1e-9, 10, 40
    ref = oracle.Multigrid(A, b, L)
    it_ref, conv_ref, rss_ref, _ = ref.solve(tol, every, iters)
    mg = amg.Multigrid.poisson(n, L, tolerance=tol, compute_error_every_n_iters=every, n_iters=iters)
    u, it, conv, last = mg.solve()
    out = capsys.readouterr().out
    assert it == it_ref and conv == conv_ref
    assert ("AMG converged after" if conv else "AMG did not converge after") in out
    assert abs(last - rss_ref) <= 1e-10 * rss_ref
    ur = ref.get_vec(0, "u")
    assert np.linalg.norm(u - ur) <= 1e-10 * np.linalg.norm(ur)
    mg.close()
