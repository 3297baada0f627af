"""Slab sharding on the device (include/amg_hip.h "slab sharding", slab_vcycle.py): the K-Patch
legs run over line ranges only, two exchanges per cycle.  Bar: the assembled solution equals
the ordinary single-GPU V-cycle bit for bit (per-row arithmetic does not depend on the
partition).  True Jacobi has no counterpart in the reference: pinned to the oracle twin
through the single-GPU cycle's own parity tests.  Nothing here reads /root/reference."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture()
def patch_everywhere(amg):
    amg.set_patch_min_rows(0)      # K-Patch on every level whose geometry allows it
    yield
    amg.set_patch_min_rows(amg.PATCH_MIN_ROWS_DEFAULT)


def _engines(amg, n, L, world, max_levels):
    sys.path.insert(0, os.path.join(ROOT, "algebraic-multigrid_amd"))
    import slab_vcycle
    dev = torch.device("cuda", 0)
    st = torch.cuda.Stream(dev)
    torch.cuda.set_stream(st)
    return [slab_vcycle.HipSlabEngine(amg, dev, st, n, L, 0.6, 2, r, world, max_levels)
            for r in range(world)], st


def _cycle(engs, poison):
    """One slab-sharded V-cycle over the engines of ALL ranks in this process: the two exchanges
    are device copies between the ranks' vectors (what RCCL does between GPUs)."""
    W = len(engs)
    i0 = engs[0].info
    m, H = int(i0.pitch0), int(i0.halo_lines) * int(i0.pitch0)
    for r, e in enumerate(engs):           # exchange 1: halo lines of u0
        a, b = int(e.info.line_begin) * m, int(e.info.line_end) * m
        if r > 0:
            e.u0[a - H:a].copy_(engs[r - 1].u0[a - H:a])
        if r < W - 1:
            e.u0[b:b + H].copy_(engs[r + 1].u0[b:b + H])
    for r, e in enumerate(engs):
        if poison and W > 1:               # nothing outside owned lines + halo may matter
            a, b = int(e.info.line_begin) * m, int(e.info.line_end) * m
            lo, hi = max(0, a - H) if r > 0 else 0, (b + H) if r < W - 1 else e.u0.numel()
            e.u0[:lo] = float("nan")
            e.u0[hi:] = float("nan")
        e.run(1)
    blk = int(i0.chunk_lines) * int(i0.gather_pitch)
    if W > 1:
        for r, e in enumerate(engs):       # exchange 2: all-gather of the replicated level's rhs
            if poison:
                keep = e.fg[r * blk:(r + 1) * blk].clone()
                e.fg[:] = float("nan")
                e.fg[r * blk:(r + 1) * blk] = keep
        for r, e in enumerate(engs):
            for q, o in enumerate(engs):
                if q != r:
                    o.fg[r * blk:(r + 1) * blk].copy_(e.fg[r * blk:(r + 1) * blk])
    for e in engs:
        e.run(2)
        e.run(3)


def _assemble(engs):
    i0 = engs[0].info
    m = int(i0.pitch0)
    n0 = int(i0.lines) * m
    out = torch.empty(n0, dtype=torch.float64, device=engs[0].u0.device)
    for e in engs:
        a, b = int(e.info.line_begin) * m, int(e.info.line_end) * m
        out[a:b] = e.u0[a:b]
    return out.cpu().numpy()


@pytest.mark.parametrize("world,max_levels", [(1, -1), (2, -1), (3, -1), (3, 2), (5, 1)])
def test_slab_cycle_equals_single_gpu_cycle(amg, patch_everywhere, world, max_levels):
    n, L, cycles = 1024, 9, 4
    ref = amg.Multigrid.poisson(n, L, smoother=amg.SM_JACOBI, smoother_iters=2, omega=0.6)
    engs, st = _engines(amg, n, L, world, max_levels)
    k = int(engs[0].info.levels)
    assert k == (4 if max_levels < 0 else max_levels)
    assert int(engs[0].info.halo_lines) == (6 * k - 1 if k > 1 else 5)
    for c in range(cycles):
        ref.vcycle()
        _cycle(engs, poison=True)
        st.synchronize()
        u = _assemble(engs)
        assert np.array_equal(u, ref.get_soln(0)), f"cycle {c}"
    assert not np.isnan(u).any() and np.abs(u).max() > 0
    for e in engs:
        e.close()
    ref.close()


@pytest.mark.parametrize("pmr,k", [(None, 5), (0, 6)])
def test_slab_full_size_4096_eight_ranks(amg, pmr, k):
    """BASELINE config 3 cut into 8 x 512 lines, all eight ranks' legs run one after the other on
    this GPU: with the single-GPU K-Patch threshold (10^6 rows: five slab levels, 29 halo lines) and the way
    bench.py --gpus 8 cuts it (--slab-patch-min-rows 0: six slab levels, 35 halo lines)."""
    n, L, world = 4096, 16, 8
    ref = amg.Multigrid.poisson(n, L, smoother=amg.SM_JACOBI, smoother_iters=2, omega=0.6)
    if pmr is not None:
        amg.set_patch_min_rows(pmr)
    try:
        engs, st = _engines(amg, n, L, world, -1)
    finally:
        amg.set_patch_min_rows(amg.PATCH_MIN_ROWS_DEFAULT)
    assert int(engs[0].info.levels) == k and int(engs[0].info.halo_lines) == 6 * k - 1
    for c in range(3):
        ref.vcycle()
        _cycle(engs, poison=True)
    st.synchronize()
    u = _assemble(engs)
    assert np.array_equal(u, ref.get_soln(0))
    for e in engs:
        e.close()
    ref.close()


def test_slab_setup_refusals(amg):
    mg = amg.Multigrid.poisson(256, 6, smoother=amg.SM_JACOBI, smoother_iters=2, omega=0.6)
    with pytest.raises(amg.AmgHipError) as ex:       # 65536 rows: below patch_min_rows, no K-Patch level
        mg.slab_setup(0, 2)
    assert ex.value.status == amg.EUNSUPPORTED
    with pytest.raises(amg.AmgHipError):
        mg.slab_run(1)
    mg.close()
    mg = amg.Multigrid.poisson(256, 6, smoother=amg.SM_SPGS)
    with pytest.raises(amg.AmgHipError):             # lexicographic GS does not shard
        mg.slab_setup(0, 2)
    mg.close()
