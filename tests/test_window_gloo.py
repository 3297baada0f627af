"""Window-sharded V-cycle on CPU (algebraic-multigrid_amd/window_vcycle.py; SURVEY 8(e), BASELINE
configs 4 and 5): every rank builds only its WINDOW of the hierarchy (the product's host setup
through the C ABI, amg_hip_create_poisson_window with host_only), runs the legs of the distributed
levels over the whole window with the oracle's arithmetic (tests/window_engine.py), exchanges the
level-0 halo and all-gathers f_k; the replicated tail is built from the all-gathered rows of A_k.
Bar: the assembled level-0 solution is BIT-EQUAL to the single-process oracle V-cycle
(multigrid.hpp:263-305) -- for true Jacobi in 2-D and 3-D and for multicolour Gauss-Seidel with
the product's colours -- under gloo (world 2, 3) and with up to 8 ranks as threads."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "algebraic-multigrid_amd"), os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

JAC, MC = 3, 4


def _reference(O, amg, dim, n, L, smoother, iters, omega, cycles):
    A, b = O.laplacian(n, dim), O.rhs(n, dim)
    kind = O.SM_TRUE_JACOBI if smoother == JAC else O.SM_MULTICOLOR
    mg = O.Multigrid(A, b, L, smoother=kind, smoother_iters=iters, omega=omega)
    if smoother == MC:   # the oracle twin replays the product's (global) greedy colouring
        h = amg.Multigrid(A.colptr, A.rowind, A.val, b, L, smoother=amg.SM_MULTICOLOR_GS, host_only=True)
        for l in range(L):
            c, nc = h.get_colors(l)
            mg.set_colors(l, c, nc)
        h.close()
    rss = []
    for _ in range(cycles):
        mg.vcycle()
        rss.append(mg.rss())
    return mg.get_vec(0, "u"), rss


def _threads(O, amg, dim, n, L, k, world, smoother, iters, omega, cycles, tamper=0):
    import window_vcycle as W
    from window_engine import EmulatedWindowEngine, ThreadComm, run_threads

    def fn(rank, hub):
        plan = W.WindowPlan(dim, n, rank, world, k, smoother, iters, tamper=tamper)
        eng = EmulatedWindowEngine(O, amg, plan, omega)
        dv = W.WindowVcycle(eng, plan, ThreadComm(hub, rank), L)
        rss = []
        for _ in range(cycles):
            dv.vcycle()
            rss.append(dv.rss())
        u = dv.gather_solution()
        chk = dv.solution_checksum()
        dv.close()
        return u, rss, chk, plan
    return run_threads(world, fn)


CASES = [
    # dim, n, L, k, world, smoother, iters, omega
    (2, 128, 7, 3, 3, JAC, 2, 0.6),     # three distributed levels, three ranks
    (2, 96, 6, 1, 4, JAC, 3, 0.5),      # odd sweep count, four ranks, ragged last block
    (2, 256, 8, 2, 8, JAC, 2, 0.6),     # eight ranks (as bench.py --gpus 8 cuts)
    (2, 70, 5, 1, 2, JAC, 2, 0.6),      # n not a power of two (70 = 2 * 35: lines of odd length below level 0)
    (3, 16, 5, 1, 2, JAC, 2, 0.6),      # 3-D 7-point: units are x-y planes
    (3, 32, 6, 2, 2, JAC, 2, 0.6),      # 3-D, two distributed levels
    (2, 128, 7, 2, 2, MC, 1, 1.0),      # multicolour GS: red-black level 0, 4 colours below
    (2, 256, 8, 3, 3, MC, 1, 1.0),
    (2, 128, 6, 1, 2, MC, 2, 1.0),      # two passes per smoothing step
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "-".join(str(x) for x in c))
def test_window_sharded_vcycle_equals_single_process(oracle, amg, case):
    dim, n, L, k, world, smoother, iters, omega = case
    cycles = 3
    res = _threads(oracle, amg, dim, n, L, k, world, smoother, iters, omega, cycles)
    u_ref, rss_ref = _reference(oracle, amg, dim, n, L, smoother, iters, omega, cycles)
    chk_ref = int(np.ascontiguousarray(u_ref).view(np.int64).sum(dtype=np.int64))
    for u, rss, chk, plan in res:
        assert np.array_equal(u, u_ref)                       # bit for bit
        assert chk == chk_ref
        for a, b in zip(rss, rss_ref):                        # partial sums follow the partition
            assert abs(a - b) <= 1e-12 * b
        if world > 1:
            assert plan.w1 - plan.w0 < n                      # a window, not the whole problem


def test_window_halo_depth_is_tight_enough_to_matter(oracle, amg):
    """With a halo three units shallower than planned the true-Jacobi cycle must NOT reproduce the
    single-process result (the plan counts whole units at the level boundaries: two units of slack
    with two distributed levels, like amg_hip_slab_plan), else the test above would prove nothing."""
    case = (2, 128, 7, 2, 2, JAC, 2, 0.6)
    u_ref, _ = _reference(oracle, amg, 2, 128, 7, JAC, 2, 0.6, 2)
    ok = _threads(oracle, amg, *case, 2, tamper=2)
    assert all(np.array_equal(r[0], u_ref) for r in ok)
    bad = _threads(oracle, amg, *case, 2, tamper=3)
    assert not any(np.array_equal(r[0], u_ref) for r in bad)


def test_window_hierarchy_equals_global_hierarchy_away_from_the_edges(oracle, amg):
    """The claim everything rests on: level matrices of a window solver (product host setup) are the
    principal submatrices of the global level matrices -- same pattern incl. structural zeros, same
    bits -- except for the rows within two units of a cut edge."""
    n, L = 64, 5
    A, b = oracle.laplacian(n), oracle.rhs(n)
    glob = amg.Multigrid(A.colptr, A.rowind, A.val, b, L, smoother=amg.SM_JACOBI, host_only=True)
    w0, w1 = 10, 40
    win = amg.Multigrid.poisson_window(n, w0, w1, L, host_only=True)
    for l in range(L):
        pitch = n >> l
        cpw, riw, vw = win.get_coefficient_matrix(l)
        cpg, rig, vg = glob.get_coefficient_matrix(l)
        nw = win.get_n_dofs(l)
        assert nw == (w1 - w0) * pitch - (1 if l else 0)
        shift = w0 * pitch
        for c in range(2 * pitch, nw - 2 * pitch):            # columns two units inside the window
            g = c + shift
            a0, a1 = cpw[c], cpw[c + 1]
            b0, b1 = cpg[g], cpg[g + 1]
            assert a1 - a0 == b1 - b0
            assert np.array_equal(riw[a0:a1] + shift, rig[b0:b1])
            assert np.array_equal(vw[a0:a1].view(np.int64), vg[b0:b1].view(np.int64))
    win.close()
    glob.close()


def test_window_plan_arithmetic():
    import window_vcycle as W
    for world in (1, 2, 5, 8):
        end = 0
        for r in range(world):
            p = W.WindowPlan(2, 4096, r, world, 4)
            assert p.own0 == end
            end = p.own1
            assert p.halo == 23                              # 6 k - 1 units, as amg_hip_slab_plan
            if world > 1:
                assert p.w0 == max(0, p.own0 - 23) and p.w1 == min(4096, p.own1 + 23)
            dlo, dhi, ulo, uhi = p.patch_ranges()
            for l in range(4):
                assert 0 <= dlo[l] <= ulo[l] <= p.own0 - p.w0 <= p.own1 - p.w0 <= uhi[l] <= dhi[l] <= p.w1 - p.w0
        assert end == 4096
    p = W.WindowPlan(2, 8192, 3, 8, 3, W.SM_MULTICOLOR, 1)    # BASELINE config 4 as bench.py cuts it
    assert p.halo == 47 and p.w0 % 2 == 0 and p.w1 - p.w0 <= 1024 + 2 * 47 + 1
    p = W.WindowPlan(3, 512, 3, 8, 2)                         # BASELINE config 5: x-y planes
    assert p.halo == 11 and p.unit_rows == 512 * 512 and p.pitch(2) == 512 * 128
    with pytest.raises(ValueError):
        W.WindowPlan(2, 64, 0, 4, 4)                         # 16 lines per rank < halo
    with pytest.raises(ValueError):
        W.WindowPlan(2, 4096, 0, 2, 2, smoother=0)           # lexicographic GS does not shard


def test_window_create_argument_errors(amg):
    with pytest.raises(ValueError):
        amg.Multigrid.poisson_window(35, 3, 20, 3, host_only=True)      # odd flat index at the window start
    with pytest.raises(ValueError):
        amg.Multigrid.poisson_window(64, 10, 70, 3, host_only=True)     # beyond the grid
    with pytest.raises(ValueError):
        amg.Multigrid.poisson_window(64, 10, 40, 1, host_only=True)     # no distributed level


# ---- the same driver under torch.distributed (gloo), one process per rank -----------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, case, cycles, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as O
    import amg_ctypes as amg
    import window_vcycle as W
    from window_engine import EmulatedWindowEngine
    dim, n, L, k, _, smoother, iters, omega = case
    plan = W.WindowPlan(dim, n, rank, world, k, smoother, iters)
    eng = EmulatedWindowEngine(O, amg, plan, omega)
    dv = W.WindowVcycle(eng, plan, W.TorchComm(rank, world), L)
    rss = []
    for _ in range(cycles):
        dv.vcycle()
        rss.append(dv.rss())
    u = dv.gather_solution()
    assert dv.solution_checksum() == int(np.ascontiguousarray(u).view(np.int64).sum(dtype=np.int64))
    if rank == world - 1:
        np.savez(os.path.join(out_dir, "win.npz"), u=u, rss=np.array(rss))
    dv.close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("case", [(2, 128, 7, 2, 2, JAC, 2, 0.6), (2, 128, 7, 2, 3, MC, 1, 1.0),
                                  (3, 16, 5, 1, 2, JAC, 2, 0.6)],
                         ids=["jacobi2d-w2", "multicolor2d-w3", "jacobi3d-w2"])
def test_window_sharded_vcycle_under_gloo(tmp_path, oracle, amg, case):
    world, cycles = case[4], 2
    mp.spawn(_worker, args=(world, _free_port(), case, cycles, str(tmp_path)), nprocs=world, join=True)
    got = np.load(os.path.join(str(tmp_path), "win.npz"))
    dim, n, L, k, _, smoother, iters, omega = case
    u_ref, rss_ref = _reference(oracle, amg, dim, n, L, smoother, iters, omega, cycles)
    assert np.array_equal(got["u"], u_ref)
    for a, b in zip(got["rss"], rss_ref):
        assert abs(a - b) <= 1e-12 * b
