"""RCCL from inside the library (include/amg_hip.h "communicator"; csrc/comm.cpp): the sharded
V-cycles as ONE C call each, the exchanges being the library's own ncclSend / ncclRecv /
ncclAllGather on the solver's stream.  A one-GPU box can only hold a communicator of ONE rank
(RCCL refuses two ranks on a device), which still runs every call of the path -- dlopen of
librccl, ncclCommInitRank, the captured legs, the staging copies, the all-gather code path of the
window cycle -- and must reproduce amg_hip_vcycle bit for bit.  The multi-rank form of the same
cycles is covered with torch.distributed / threads in tests/test_window_gloo.py,
tests/test_gpu_window.py, tests/test_dist_gloo.py, tests/test_gpu_slab.py."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (os.path.join(ROOT, "algebraic-multigrid_amd"), os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)


@pytest.fixture()
def comm1(amg):
    c = amg.Comm(amg.Comm.unique_id(), 0, 1)
    yield c
    c.close()


def test_slab_cycle_through_the_library_communicator(amg, comm1):
    n, L = 1024, 10
    amg.set_patch_min_rows(0)
    try:
        ref = amg.Multigrid.poisson(n, L, smoother=amg.SM_JACOBI, smoother_iters=2, omega=0.6)
        mg = amg.Multigrid.poisson(n, L, smoother=amg.SM_JACOBI, smoother_iters=2, omega=0.6)
    finally:
        amg.set_patch_min_rows(amg.PATCH_MIN_ROWS_DEFAULT)
    info = mg.slab_setup(0, 1)
    assert info.levels >= 3
    for c in range(3):
        ref.vcycle()
        comm1.slab_cycle(mg, info)
        mg.sync()
        assert np.array_equal(mg.get_soln(0), ref.get_soln(0)), c
    assert mg.rss() == ref.rss()
    mg.close()
    ref.close()


@pytest.mark.parametrize("dim,n,L,k,smoother", [(2, 512, 9, 3, "jacobi"), (2, 512, 9, 2, "multicolor"),
                                                (3, 48, 7, 2, "jacobi")])
def test_window_cycle_through_the_library_communicator(amg, comm1, dim, n, L, k, smoother):
    import window_vcycle as W
    from window_engine import ThreadComm, ThreadHub
    sm_w = W.SM_JACOBI if smoother == "jacobi" else W.SM_MULTICOLOR
    iters = 2 if smoother == "jacobi" else 1
    omega = 0.6 if smoother == "jacobi" else 1.0
    dev = torch.device("cuda", 0)
    st = torch.cuda.Stream(dev)
    torch.cuda.set_stream(st)
    plan = W.WindowPlan(dim, n, 0, 1, k, sm_w, iters)
    eng = W.HipWindowEngine(amg, dev, st, plan, omega)
    dv = W.WindowVcycle(eng, plan, ThreadComm(ThreadHub(1), 0, sync=st.synchronize), L)
    dv.use_library_comm(comm1)
    sm = amg.SM_JACOBI if smoother == "jacobi" else amg.SM_MULTICOLOR_GS
    ref = amg.Multigrid.poisson(n, L, dim=dim, smoother=sm, smoother_iters=iters, omega=omega)
    for c in range(3):
        ref.vcycle()
        dv.vcycle()
    assert np.array_equal(dv.gather_solution(), ref.get_soln(0))
    assert abs(dv.rss() - ref.rss()) <= 1e-12 * ref.rss()
    dv.close()
    ref.close()
    torch.cuda.set_stream(torch.cuda.default_stream(dev))


def test_comm_argument_errors(amg):
    with pytest.raises(amg.AmgHipError):
        amg.Comm(amg.Comm.unique_id(), 3, 2)        # rank outside the world
