"""CPU-side checks of the drop-in boundary: the shared library loads and exports
every symbol include/amg_hip.h declares; argument validation that needs no GPU."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "amg_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(amg_hip_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_something():
    syms = declared_symbols()
    assert len(syms) >= 30 and "amg_hip_vcycle" in syms


def test_library_exports_every_declared_symbol(amg):
    L = ctypes.CDLL(amg.LIB_PATH)
    missing = [s for s in declared_symbols() if not hasattr(L, s)]
    assert not missing, missing


def test_binding_table_matches_header(amg):
    assert sorted(amg._SIGS) == declared_symbols()


def test_generators_run_on_host(amg, oracle):
    # Grid<double> restatement in the product == oracle, bit for bit
    for n in (2, 35, 128):
        cp, ri, v = amg.laplacian(n)
        A = oracle.laplacian(n)
        assert (cp == A.colptr).all() and (ri == A.rowind).all() and (v == A.val).all()
        assert (amg.rhs(n) == oracle.rhs(n)).all()
    cp, ri, v = amg.laplacian(9, dim=3)
    A = oracle.laplacian(9, dim=3)
    assert (cp == A.colptr).all() and (ri == A.rowind).all() and (v == A.val).all()
    assert (amg.rhs(9, dim=3) == oracle.rhs(9, dim=3)).all()


def test_ctor_argument_errors_match_reference(amg, oracle):
    # testlib.cpp:131-144 -> std::invalid_argument (ValueError here), same text
    A, b = oracle.laplacian(2), oracle.rhs(2)
    with pytest.raises(ValueError, match="`compute_error_every_n_iters` must be leq to `n_iters`, got 100 and 10"):
        amg.Multigrid(A.colptr, A.rowind, A.val, b, 8, compute_error_every_n_iters=100, n_iters=10)
    import numpy as np
    bad_colptr = np.zeros(11, np.int32)
    with pytest.raises(ValueError, match="same number of degrees of freedom, got 10 and 11"):
        amg.Multigrid(bad_colptr, np.zeros(0, np.int32), np.zeros(0), np.zeros(11), 8,
                      compute_error_every_n_iters=10, n_iters=100)


def test_no_cpu_fallback_without_device(amg, oracle):
    """On a box without a GPU every compute entry point must fail loudly."""
    if amg.device_count() > 0:
        pytest.skip("a HIP device is present")
    A, b = oracle.laplacian(4), oracle.rhs(4)
    with pytest.raises(amg.AmgHipError) as e:
        amg.residual(A.colptr, A.rowind, A.val, b, b)
    assert e.value.status == amg.EHIP
    with pytest.raises(amg.AmgHipError) as e:
        amg.Multigrid(A.colptr, A.rowind, A.val, b, 2)
    assert e.value.status == amg.EHIP


def test_dictionary_coding_round_trips_on_the_host(amg, oracle):
    """K-Dict's encoder (host side of the default device layout): every level of real
    hierarchies qualifies, holds the handful of pairs / row types DESIGN.md quotes, and
    decodes back to the exact input; matrices that cannot qualify are refused."""
    import numpy as np
    for n, dim, L in ((96, 2, 6), (12, 3, 5)):
        A, b = oracle.laplacian(n, dim=dim), oracle.rhs(n, dim=dim)
        mg = oracle.Multigrid(A, b, L)
        for l in range(L):
            M = mg.level_matrix(l)            # symmetric: CSC arrays == CSR arrays
            got = amg.dict_probe(M.colptr, M.rowind, M.val, M.cols)
            assert got is not None, (n, dim, l)
            pairs, types, words = got
            assert 1 <= pairs <= 64 and 1 <= types <= 64 and words in (1, 2), (n, dim, l, got)
    cp, ri, v = amg.laplacian(64)
    assert amg.dict_probe(cp, ri, v, 64 * 64) == (5, 9, 1)     # 2-D 5-point: 5 pairs, 9 distinct rows
    cp, ri, v = amg.laplacian(1024)                             # >= 2^20 rows: the threaded encoder
    assert amg.dict_probe(cp, ri, v, 1024 * 1024) == (5, 9, 1)
    # a halo-extended local block (multi-GPU): columns shifted by the halo width
    rows = 40
    rp = np.arange(0, 3 * rows + 1, 3, dtype=np.int32)
    col = (np.arange(rows)[:, None] + np.array([0, 7, 14])[None, :]).astype(np.int32).ravel()
    val = np.tile(np.array([-1.0, 4.0, -1.0]), rows)
    assert amg.dict_probe(rp, col, val, rows + 14, diag_shift=7) == (3, 1, 1)
    # more than 255 distinct values: refused (SELL-64 is used instead)
    n = 400
    rp = np.arange(n + 1, dtype=np.int32)
    assert amg.dict_probe(rp, np.arange(n, dtype=np.int32), np.arange(1.0, n + 1.0), n) is None
    # a row of 17 entries: refused
    rp = np.array([0, 17], dtype=np.int32)
    assert amg.dict_probe(rp, np.arange(17, dtype=np.int32), np.ones(17), 17) is None
    # rows with more than 255 distinct code words but few pairs: first level only
    rng = np.random.default_rng(3)
    n = 3000
    cols, vals, rp = [], [], [0]
    for i in range(n):
        pick = sorted(set(int(c) for c in i + rng.integers(-3, 4, size=4) if 0 <= c < n) | {i})
        cols += pick
        vals += [4.0 if c == i else -1.0 for c in pick]
        rp.append(len(cols))
    got = amg.dict_probe(np.array(rp, np.int32), np.array(cols, np.int32), np.array(vals), n)
    assert got is not None and got[0] <= 14 and got[2] == 1
