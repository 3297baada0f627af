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
