"""Edge cases of the C ABI on the GPU: smallest sizes, single level, ragged /
non-symmetric inputs, error contracts (SURVEY 8(b) 'Errors')."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def csc(A):
    return A.colptr, A.rowind, A.val


def test_single_level_is_a_direct_solve(amg, oracle):
    # n_levels = 1: smooth + residual on the only level, then the band solve overwrites u
    A, b = oracle.laplacian(6), oracle.rhs(6)
    ref = oracle.Multigrid(A, b, 1)
    mg = amg.Multigrid(*csc(A), b, 1)
    ref.vcycle()
    mg.vcycle()
    u = mg.get_soln(0)
    assert np.array_equal(u, ref.get_vec(0, "u"))
    exact = np.linalg.solve(A.to_scipy().toarray(), b)
    assert np.linalg.norm(u - exact) <= 1e-12 * np.linalg.norm(exact)
    mg.close()


def test_smallest_problems(amg, oracle):
    # 1x1 grid (one dof), 2x2 grid with two levels (coarse level has 1 dof)
    A, b = oracle.laplacian(1), oracle.rhs(1)
    mg = amg.Multigrid(*csc(A), b, 1)
    mg.vcycle()
    assert np.array_equal(mg.get_soln(0), b / A.val[0])
    mg.close()
    A, b = oracle.laplacian(2), oracle.rhs(2)
    ref = oracle.Multigrid(A, b, 2)
    mg = amg.Multigrid(*csc(A), b, 2)
    assert [mg.get_n_dofs(0), mg.get_n_dofs(1)] == [4, 1]
    for _ in range(3):
        ref.vcycle()
        mg.vcycle()
    assert np.array_equal(mg.get_soln(0), ref.get_vec(0, "u"))
    mg.close()


def test_too_many_levels_is_an_argument_error(amg, oracle):
    # 35^2 supports 9 levels (…, 8, 3, 1): a 11th would be empty; the reference has UB here
    A, b = oracle.laplacian(35), oracle.rhs(35)
    with pytest.raises(ValueError, match="no degrees of freedom"):
        amg.Multigrid(*csc(A), b, 12)


def test_malformed_matrices_are_rejected(amg, oracle):
    A, b = oracle.laplacian(4), oracle.rhs(4)
    bad = A.rowind.copy()
    bad[1], bad[0] = bad[0], bad[1]                      # unsorted column
    with pytest.raises(ValueError, match="ascending"):
        amg.Multigrid(A.colptr, bad, A.val, b, 2)
    bad = A.rowind.copy()
    bad[3] = 999                                          # index out of range
    with pytest.raises(ValueError, match="out of range"):
        amg.Multigrid(A.colptr, bad, A.val, b, 2)
    with pytest.raises((ValueError, amg.AmgHipError)):
        amg.residual(A.colptr, bad, A.val, b, b)


def test_solve_argument_check(amg, oracle):
    A, b = oracle.laplacian(8), oracle.rhs(8)
    with pytest.raises(ValueError, match="must be leq to `n_iters`, got 100 and 10"):
        amg.Multigrid(*csc(A), b, 2, compute_error_every_n_iters=100, n_iters=10)


def test_nonsymmetric_matrix_column_as_row_semantics(amg, oracle):
    """SpGS and the build-side smoothers walk COLUMN c of the CSC as if it
    were row c (A^T), while the residual uses A.  Reproduced exactly."""
    import scipy.sparse as sp
    n = 400
    rng = np.random.default_rng(3)
    S = sp.diags([-1.0, 4.0, -1.3], [-1, 0, 1], shape=(n, n), format="csc")
    S = S + sp.random(n, n, density=0.004, random_state=5, format="csc") * 0.1
    S = sp.csc_matrix(S)
    S.sort_indices()
    A = oracle.CSC(n, n, S.indptr, S.indices, S.data)
    u0, b = rng.standard_normal(n), rng.standard_normal(n)
    for kind, okind in ((amg.SM_SPGS, oracle.SM_SPGS), (amg.SM_JACOBI, oracle.SM_TRUE_JACOBI),
                        (amg.SM_SOR, oracle.SM_SOR), (amg.SM_REF_JACOBI, oracle.SM_REF_JACOBI)):
        got, _, _ = amg.smooth(kind, *csc(A), u0, b, n_iters=2, omega=0.9, every=100)
        want, _, _ = oracle.smooth(okind, A, u0, b, n_iters=2, omega=0.9, every=100)
        assert np.array_equal(got, want), kind
    assert np.array_equal(amg.residual(*csc(A), u0, b), oracle.residual(A, u0, b))


def test_rows_without_diagonal_are_left_alone(amg, oracle):
    # smoother.hpp:136: diag == 0 -> u[col] unchanged
    import scipy.sparse as sp
    n = 50
    S = sp.diags([1.0, -3.0, 1.0], [-1, 0, 1], shape=(n, n), format="lil")
    S[7, 7] = 0.0
    S[20, 20] = 0.0
    S = S.tocsc()
    S.eliminate_zeros()
    S.sort_indices()
    A = oracle.CSC(n, n, S.indptr, S.indices, S.data)
    rng = np.random.default_rng(1)
    u0, b = rng.standard_normal(n), rng.standard_normal(n)
    for kind, okind in ((amg.SM_SPGS, oracle.SM_SPGS), (amg.SM_JACOBI, oracle.SM_TRUE_JACOBI)):
        got, _, _ = amg.smooth(kind, *csc(A), u0, b, n_iters=3, omega=0.8)
        want, _, _ = oracle.smooth(okind, A, u0, b, n_iters=3, omega=0.8)
        assert np.array_equal(got, want)
        assert got[7] == u0[7] and got[20] == u0[20]


def test_two_solvers_are_independent(amg, oracle):
    A1, b1 = oracle.laplacian(20), oracle.rhs(20)
    A2, b2 = oracle.laplacian(31), oracle.rhs(31)
    r1, r2 = oracle.Multigrid(A1, b1, 3), oracle.Multigrid(A2, b2, 4)
    m1, m2 = amg.Multigrid(*csc(A1), b1, 3), amg.Multigrid(*csc(A2), b2, 4)
    for _ in range(3):
        m1.vcycle()
        m2.vcycle()
        r1.vcycle()
        r2.vcycle()
    assert np.array_equal(m1.get_soln(0), r1.get_vec(0, "u"))
    assert np.array_equal(m2.get_soln(0), r2.get_vec(0, "u"))
    m1.close()
    m2.close()


def test_device_pointer_csr_api(amg, oracle):
    """amg_hip_dev_* launchers on caller-owned device memory (torch tensors): plain CSR
    arrays, halo-style column offset (diag_shift), explicit stream."""
    import ctypes as C
    import torch
    A = oracle.laplacian(40)
    n = A.rows
    rng = np.random.default_rng(4)
    u, f = rng.standard_normal(n), rng.standard_normal(n)
    lib = amg.lib()
    dev = torch.device("cuda", 0)
    mb, mr = amg.csr_shape(A.colptr)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    rp, ci, va = t(A.colptr), t(A.rowind), t(A.val)          # symmetric: CSC == CSR
    # vector extended by a fake halo of 7 entries in front: columns shift by 7
    shift = 7
    ci_s = t(A.rowind + shift)
    u_ext = t(np.concatenate([np.full(shift, 1e300), u]))
    ft, out = t(f), torch.empty(n, dtype=torch.float64, device=dev)
    st = torch.cuda.Stream(dev)
    with torch.cuda.stream(st):
        assert lib.amg_hip_dev_residual(n, A.nnz, mb, mr, rp.data_ptr(), ci_s.data_ptr(), va.data_ptr(),
                                        u_ext.data_ptr(), ft.data_ptr(), out.data_ptr(), st.cuda_stream) == 0
        st.synchronize()
        assert np.array_equal(out.cpu().numpy(), oracle.residual(A, u, f))
        assert lib.amg_hip_dev_jacobi(n, A.nnz, mb, mr, rp.data_ptr(), ci_s.data_ptr(), va.data_ptr(),
                                      u_ext.data_ptr(), ft.data_ptr(), out.data_ptr(), 0.7, shift,
                                      st.cuda_stream) == 0
        st.synchronize()
        want, _, _ = oracle.smooth(oracle.SM_TRUE_JACOBI, A, u, f, n_iters=1, omega=0.7)
        assert np.array_equal(out.cpu().numpy(), want)
        assert lib.amg_hip_dev_spmv(n, A.nnz, mb, mr, rp.data_ptr(), ci.data_ptr(), va.data_ptr(),
                                    t(u).data_ptr(), out.data_ptr(), st.cuda_stream) == 0
        st.synchronize()
        assert np.array_equal(out.cpu().numpy(), oracle.spmv(A, u))
        # from-zero sweep == full sweep on a zero vector
        diag = t(A.to_scipy().diagonal())
        assert lib.amg_hip_dev_jacobi_from_zero(n, diag.data_ptr(), ft.data_ptr(), out.data_ptr(), 0.7,
                                                st.cuda_stream) == 0
        st.synchronize()
        want, _, _ = oracle.smooth(oracle.SM_TRUE_JACOBI, A, np.zeros(n), f, n_iters=1, omega=0.7)
        assert np.array_equal(out.cpu().numpy(), want)
        # y += x and sum of squares
        y = t(u.copy())
        assert lib.amg_hip_dev_axpy1(n, ft.data_ptr(), y.data_ptr(), st.cuda_stream) == 0
        scratch = torch.zeros(1100, dtype=torch.float64, device=dev)
        assert lib.amg_hip_dev_sumsq(n, ft.data_ptr(), scratch[1024:].data_ptr(), scratch.data_ptr(),
                                     st.cuda_stream) == 0
        st.synchronize()
        assert np.array_equal(y.cpu().numpy(), u + f)
        assert abs(float(scratch[1024]) - float(np.dot(f, f))) <= 1e-13 * np.dot(f, f)
    # misaligned matrix arrays are refused, not mis-read
    assert lib.amg_hip_dev_spmv(n, A.nnz, mb, mr, rp.data_ptr(), ci.data_ptr() + 4, va.data_ptr(),
                                t(u).data_ptr(), out.data_ptr(), None) == amg.EINVAL


def test_non_finite_input_ends_the_solve_loop_like_the_reference(amg, oracle, capsys):
    """NaN in b: the reference's loop `while (iter < n_iters && error > tolerance)`
    (multigrid.hpp:317) leaves at the first rss check because NaN > tol is false, and prints
    "did not converge".  Same here: no hang, rss is NaN, converged is False.  (Which ENTRIES turn
    NaN can differ from the reference: the device kernels skip exact-zero entries and unused
    slots, x + 0*NaN is only defined for finite vectors -- outside the parity contract.)"""
    n, L = 24, 3
    A, b = oracle.laplacian(n), oracle.rhs(n)
    b = b.copy()
    b[100] = np.nan
    for kw in (dict(), dict(smoother=amg.SM_JACOBI, smoother_iters=2, omega=0.6)):
        mg = amg.Multigrid(*csc(A), b, L, compute_error_every_n_iters=5, n_iters=100, **kw)
        u, iters, converged, last = mg.solve()
        assert iters == 5 and not converged and np.isnan(last)
        assert np.isnan(mg.rss()) and np.isnan(u).any()
        mg.close()
    assert "AMG did not converge after 5 iterations." in capsys.readouterr().out
