#!/usr/bin/env python3
"""Compute-only time of one rank's share of a slab-sharded V-cycle (slab_vcycle.py), measured on
ONE GPU: the three graph launches of amg_hip_slab_run for a middle rank of `world`, without the
two exchanges.  An upper bound for the strong scaling of bench.py --gpus N (what RCCL adds on top
cannot be measured on a one-GPU box).  usage: slab_estimate.py [n] [levels]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "algebraic-multigrid_amd"))
sys.path.insert(0, ROOT)
import amg_ctypes as amg  # noqa: E402
import slab_vcycle  # noqa: E402
from bench import n_levels_for  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
L = int(sys.argv[2]) if len(sys.argv) > 2 else n_levels_for(n)
dev = torch.device("cuda", 0)
st = torch.cuda.Stream(dev)
torch.cuda.set_stream(st)
mg = amg.Multigrid.poisson(n, L, smoother=amg.SM_JACOBI, smoother_iters=2, omega=0.6, stream=st.cuda_stream)
mg.vcycle(10)
mg.sync()
t = time.perf_counter()
mg.vcycle(100)
mg.sync()
whole = (time.perf_counter() - t) / 100
print(f"n={n} levels={L}: whole cycle (one graph) {whole*1e3:.3f} ms", flush=True)
mg.close()
pmr = int(sys.argv[3]) if len(sys.argv) > 3 else None
for world in (1, 2, 4, 8):
    for k in ((-1, 3, 2) if pmr is None else (6, 5, 4)):
        e = slab_vcycle.HipSlabEngine(amg, dev, st, n, L, 0.6, 2, world // 2, world, k, patch_min_rows=pmr)
        parts = []
        for part in (1, 2, 3):
            for _ in range(5):
                e.run(part)
            e.sync()
            t = time.perf_counter()
            for _ in range(50):
                e.run(part)
            e.sync()
            parts.append((time.perf_counter() - t) / 50)
        tot = sum(parts)
        i = e.info
        print(f"world={world} slab levels={i.levels} halo={i.halo_lines} lines: down {parts[0]*1e6:.0f} us, "
              f"replicated rest {parts[1]*1e6:.0f} us, up {parts[2]*1e6:.0f} us = {tot*1e3:.3f} ms "
              f"({whole/tot:.2f}x the whole cycle; exchanges not included: "
              f"{2*i.halo_lines*i.pitch0*8/1e3:.0f} KB of halo lines, all-gather of {i.gather_rows*8/1e6:.1f} MB)",
              flush=True)
        e.close()
