"""Round-3 GPU parity tests of the multicolour smoother's patch form on EVERY colouring the 2-D
Poisson hierarchy produces (kernels.hip: patch_rb_kernel with a stage list and a 2 x 2 colour
table): the checkerboard of the 5-point fine level, line parity on level 1 (whose +-1 entries are
exact zeros), the 4-colour product colouring of the 9-point levels.

Parity note: the multicolour smoother has no counterpart in the reference; it is
pinned bit for bit to the oracle twin (which replays the solver's colours) and to the literal
sequence of one launch per colour (no_fusion).  Nothing here reads /root/reference."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def csc(A):
    return A.colptr, A.rowind, A.val


@pytest.fixture
def patch_everywhere(amg):
    amg.set_patch_min_rows(0)
    yield
    amg.set_patch_min_rows(amg.PATCH_MIN_ROWS_DEFAULT)


@pytest.mark.parametrize("iters", [1, 2, 3])
def test_multicolor_patch_stages_equal_the_literal_colour_sequence(amg, patch_everywhere, iters):
    """1024^2 / 7 levels, non-zero start, K-Patch threshold 0: levels 0-3 (pitch 1024 .. 128) run the
    symmetric passes as patch stages -- 2, 2 (by lines) and 4 colours; a colour that directly follows
    itself dropped; up to 3 stages per launch, the level vector travelling u -> tmp -> r -> u --
    against one launch per colour in the literal order 0 .. nc-1, nc-1 .. 0 per pass (no_fusion):
    every level vector bitwise after 3 cycles."""
    n, L = 1024, 7
    cp, ri, v = amg.laplacian(n)
    b = amg.rhs(n)
    u0 = np.random.default_rng(11).standard_normal(n * n)
    out = []
    for nf in (False, True):
        mg = amg.Multigrid(cp, ri, v, b, L, smoother=amg.SM_MULTICOLOR_GS, smoother_iters=iters, no_fusion=nf)
        assert mg.fine_sweep_info()[0].startswith("patch_rb_kernel") != nf
        if not nf:
            ncs = [mg.get_colors(l)[1] for l in range(4)]
            assert ncs[0] == 2 and ncs[2:] == [4, 4] and ncs[1] in (2, 4), ncs
        mg.set_vec(0, "u", u0)
        mg.vcycle(3)
        out.append(([mg.get_soln(l) for l in range(L)], [mg.get_rhs(l) for l in range(1, L)], mg.rss()))
        mg.close()
    for l in range(L):
        assert np.array_equal(out[0][0][l], out[1][0][l]), l
    for l in range(L - 1):
        assert np.array_equal(out[0][1][l], out[1][1][l]), l + 1
    assert out[0][2] == out[1][2]


@pytest.mark.parametrize("iters", [1, 2])
def test_multicolor_patch_stages_against_the_oracle_512(amg, oracle, patch_everywhere, iters):
    """512^2 / 6 levels with the residual kept (the third vector of the stage chain is then a
    buffer of its own): u, f and r of every level against the oracle twin after every cycle."""
    n, L = 512, 6
    A, b = oracle.laplacian(n), oracle.rhs(n)
    mg = amg.Multigrid(*csc(A), b, L, smoother=amg.SM_MULTICOLOR_GS, smoother_iters=iters,
                       exact_coarse_solve=True, keep_residual=True)
    assert mg.fine_sweep_info()[0].startswith("patch_rb_kernel")
    ref = oracle.Multigrid(A, b, L, smoother=oracle.SM_MULTICOLOR, smoother_iters=iters)
    for l in range(L):
        col, nc = mg.get_colors(l)
        ref.set_colors(l, col, nc)
    for c in range(2):
        ref.vcycle()
        mg.vcycle()
        for l in range(L):
            if l < L - 1:
                assert np.array_equal(mg.get_soln(l), ref.get_vec(l, "u")), (c, l)
                assert np.array_equal(mg.get_residual(l), ref.get_vec(l, "r")), (c, l)
            assert np.array_equal(mg.get_rhs(l), ref.get_vec(l, "f")), (c, l)
    mg.close()


def test_multicolor_default_threshold_uses_patch_stages_below_level_0(amg):
    """2048^2 / 9 levels at the default threshold (10^6 rows): levels 0-2 take the patch form; same
    bits as the literal colour sequence, and the cycle has less to move: 6 launches for a 4-colour
    level (7 stages: [3] + [2] + [2 with the restriction] down, [3] + [2] + [2] up) where the colour
    kernels stream every row 8 times per leg."""
    n, L = 2048, 9
    cp, ri, v = amg.laplacian(n)
    b = amg.rhs(n)
    res = []
    for nf in (False, True):
        mg = amg.Multigrid(cp, ri, v, b, L, smoother=amg.SM_MULTICOLOR_GS, smoother_iters=1, no_fusion=nf)
        mg.vcycle(4)
        res.append((mg.get_soln(0), mg.get_soln(2), mg.get_soln(3), mg.rss(), mg.cycle_must_move()))
        mg.close()
    for a, c in zip(res[0][:3], res[1][:3]):
        assert np.array_equal(a, c)
    assert res[0][3] == res[1][3]
    assert res[0][4] < 0.8 * res[1][4]


def test_3d_stencil_rows_paired_loads_are_bit_neutral(amg):
    """3-D 7-point 160^3 / 13 levels (4.1 M rows; lines of 160, so most waves of 128 rows lie inside
    one line and share the interior row type): the paired-load path of wave-uniform 7- / 15-point
    rows (dict_rows_stencil) and the slab-per-plane tile order against plain dict_rows in the
    round-2 tile order -- every level vector bitwise after 3 cycles, same rss."""
    n, L = 160, 13
    out = []
    for on in (1, 0):
        amg.set_dict_stencil(on)
        amg.set_xcd_mapping(1 if on else 2)
        try:
            mg = amg.Multigrid.poisson(n, L, dim=3, smoother=amg.SM_JACOBI, smoother_iters=2, omega=0.6)
            mg.vcycle(3)
            out.append(([mg.get_soln(l) for l in range(L)], [mg.get_rhs(l) for l in range(1, L)], mg.rss()))
            mg.close()
        finally:
            amg.set_dict_stencil(1)
            amg.set_xcd_mapping(1)
    for l in range(L):
        assert np.array_equal(out[0][0][l], out[1][0][l]), l
    for l in range(L - 1):
        assert np.array_equal(out[0][1][l], out[1][1][l]), l + 1
    assert out[0][2] == out[1][2]


@pytest.mark.parametrize("n,L,kind", [(2048, 5, "spgs"), (3001, 8, "spgs"), (2050, 6, "sor")])
def test_scan_long_chunks_against_the_oracle(amg, oracle, n, L, kind):
    """K-GS-scan with chunks of more than 1024 rows (grid lines of 2048 / 4096+: 2 or 4 consecutive rows
    per thread, gs_scan_typed_rows_kernel) -- the reference's own sweeps (smoother.hpp:148-174,
    :339-372) against the sequential oracle: 1e-10 on every level after 2 cycles, incl. ragged line
    ends (n odd) and lines that are not a multiple of the rows per thread."""
    A, b = oracle.laplacian(n), oracle.rhs(n)
    kw_o = dict(smoother=oracle.SM_SOR, smoother_iters=1, omega=1.3) if kind == "sor" else dict(smoother=oracle.SM_SPGS)
    kw_g = dict(smoother=amg.SM_SOR, smoother_iters=1, omega=1.3) if kind == "sor" else dict(smoother=amg.SM_SPGS)
    ref = oracle.Multigrid(A, b, L, **kw_o)
    mg = amg.Multigrid(*csc(A), b, L, **kw_g)
    for c in range(2):
        ref.vcycle()
        mg.vcycle()
    for l in range(L):
        ur = ref.get_vec(l, "u")
        assert np.linalg.norm(mg.get_soln(l) - ur) <= 1e-10 * np.linalg.norm(ur), (kind, l)
    assert abs(mg.rss() - ref.rss()) <= 1e-10 * ref.rss()
    mg.close()


@pytest.mark.parametrize("n,L", [(1024, 12), (600, 10)])
def test_scan_on_the_deep_levels_against_the_oracle(amg, oracle, n, L):
    """The reference's SparseGaussSeidel() on a deep hierarchy: the levels whose grid lines are down
    to 4 / 2 / 1 columns (chunks of 3 and 2 rows, one wave each) take K-GS-scan too since round 3
    (they were the exact kernel's) -- 1e-10 against the sequential oracle on every level after 2
    cycles, through both constructors (device-only setup and host arrays)."""
    A, b = oracle.laplacian(n), oracle.rhs(n)
    ref = oracle.Multigrid(A, b, L)
    mgs = [amg.Multigrid(*csc(A), b, L), amg.Multigrid.poisson(n, L)]
    for _ in range(2):
        ref.vcycle()
        for mg in mgs:
            mg.vcycle()
    for mg in mgs:
        for l in range(L):
            ur = ref.get_vec(l, "u")
            assert np.linalg.norm(mg.get_soln(l) - ur) <= 1e-10 * np.linalg.norm(ur), l
        assert abs(mg.rss() - ref.rss()) <= 1e-10 * ref.rss()
        mg.close()


@pytest.mark.parametrize("n,L", [(64, 9), (128, 11)])
def test_march_two_sweeps_in_one_pass_3d(amg, oracle, n, L):
    """K-March (round 3): the two plain Jacobi sweeps of the 3-D 7-point fine level as ONE plane-marching
    launch (down-leg and up-leg; the level's two vectors trade places after each) against two
    launches of the dictionary sweep (no_fusion) -- every level vector bitwise after 3 cycles --
    and, at 64^3, against the oracle twin."""
    kw = dict(smoother=amg.SM_JACOBI, smoother_iters=2, omega=0.6)
    out = []
    for nf in (False, True):
        mg = amg.Multigrid.poisson(n, L, dim=3, no_fusion=nf, **kw)
        mg.vcycle(3)
        out.append(([mg.get_soln(l) for l in range(L)], [mg.get_rhs(l) for l in range(1, L)], mg.rss()))
        if not nf and n == 64:
            A, b = oracle.laplacian(n, dim=3), oracle.rhs(n, dim=3)
            ref = oracle.Multigrid(A, b, L, smoother=oracle.SM_TRUE_JACOBI, smoother_iters=2, omega=0.6)
            for _ in range(3):
                ref.vcycle()
            assert np.array_equal(out[0][0][0], ref.get_vec(0, "u"))
            assert np.array_equal(out[0][0][1], ref.get_vec(1, "u"))
        mg.close()
    for l in range(L):
        assert np.array_equal(out[0][0][l], out[1][0][l]), l
    for l in range(L - 1):
        assert np.array_equal(out[0][1][l], out[1][1][l]), l + 1
    assert out[0][2] == out[1][2]
