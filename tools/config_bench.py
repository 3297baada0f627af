#!/usr/bin/env python3
"""Times one V-cycle configuration on the GPU (BASELINE.json configs other than
the bench.py headline).  usage: config_bench.py <dim> <n> <levels> <smoother> [cycles]
smoother: spgs | jacobi | multicolor"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "algebraic-multigrid_amd"))
import amg_ctypes as amg  # noqa: E402

dim, n, L, sm = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
cycles = int(sys.argv[5]) if len(sys.argv) > 5 else 10
fast = len(sys.argv) > 6 and sys.argv[6] == "fast"
t0 = time.time()
cp, ri, v = amg.laplacian(n, dim)
b = amg.rhs(n, dim)
kw = {"spgs": dict(smoother=amg.SM_SPGS, smoother_iters=1),
      "jacobi": dict(smoother=amg.SM_JACOBI, smoother_iters=2, omega=0.6),
      "multicolor": dict(smoother=amg.SM_MULTICOLOR_GS, smoother_iters=1)}[sm]
mg = amg.Multigrid(cp, ri, v, b, L, fast_coarse_solve=fast, **kw)
setup = time.time() - t0
mg.vcycle(2)
mg.sync()
r0 = mg.rss()
t1 = time.perf_counter()
mg.vcycle(cycles)
mg.sync()
dt = (time.perf_counter() - t1) / cycles
print(f"dim={dim} n={n} dofs={n**dim} levels={L} smoother={sm}{' fast-coarse' if fast else ''}: setup {setup:.1f}s, "
      f"{dt*1e3:.3f} ms/V-cycle = {1/dt:.2f} V-cycles/s, coarsest {mg.get_n_dofs(L-1)} dofs "
      f"(half-bw {mg.coarse_halfbw()}), rss {r0:.4e} -> {mg.rss():.4e}", flush=True)
