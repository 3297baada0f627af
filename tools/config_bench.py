#!/usr/bin/env python3
"""Times V-cycle configurations on one GPU (the BASELINE.json configs other than the bench.py
headline) and records the convergence factor next to every rate.
usage: config_bench.py <dim> <n> <levels> <smoother> [cycles]     one configuration
       config_bench.py all                                         the README table
smoother: spgs | jacobi | multicolor.  Setup runs on the device (amg_hip_create_poisson);
smoothers that need host structures fall back to the host path inside it."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "algebraic-multigrid_amd"))
import amg_ctypes as amg  # noqa: E402

KW = {"spgs": dict(smoother=amg.SM_SPGS, smoother_iters=1),
      "jacobi": dict(smoother=amg.SM_JACOBI, smoother_iters=2, omega=0.6),
      "jacobi1": dict(smoother=amg.SM_JACOBI, smoother_iters=1, omega=0.6),
      "multicolor": dict(smoother=amg.SM_MULTICOLOR_GS, smoother_iters=1)}


def run(dim, n, L, sm, cycles=10, warm=6, **extra):
    t0 = time.time()
    mg = amg.Multigrid.poisson(n, L, dim=dim, **KW[sm], **extra)
    mg.sync()
    setup = time.time() - t0
    mg.vcycle(warm)
    mg.sync()
    r0 = mg.rss()
    t1 = time.perf_counter()
    mg.vcycle(cycles)
    mg.sync()
    dt = (time.perf_counter() - t1) / cycles
    r1 = mg.rss()
    fac = (r1 / r0) ** (0.5 / cycles) if r0 > 0 and r1 > 0 else float("nan")   # per cycle, residual 2-norm
    tag = " ".join(f"{k}={v}" for k, v in extra.items())
    print(f"dim={dim} n={n} dofs={n**dim} levels={L} smoother={sm} {tag}: setup {setup:.2f}s, "
          f"{dt*1e3:.3f} ms/V-cycle = {1/dt:.1f} V-cycles/s, coarsest {mg.get_n_dofs(L-1)} dofs "
          f"(half-bw {mg.coarse_halfbw()}, {mg.coarse_solve_kind().split(' ')[0]}), "
          f"||r|| factor per cycle {fac:.4f} (rss {r0:.4e} -> {r1:.4e} over {cycles} cycles after {warm})",
          flush=True)
    mg.close()


def run_rs(n, sm, cycles=10, theta=0.25, min_coarse=500, dim=2):
    """Strength-based C/F hierarchy (amg_hip_create_rs) on the same problem; reports the cycles
    and time to reduce the residual norm by 1e-8 next to the rate."""
    cp, ri, v = amg.laplacian(n, dim)
    b = amg.rhs(n, dim)
    t0 = time.time()
    mg = amg.Multigrid.ruge_stueben(cp, ri, v, b, 25, theta, min_coarse, **KW[sm])
    mg.sync()
    setup = time.time() - t0
    L = mg.n_levels
    r0 = mg.rss()
    t1 = time.perf_counter()
    mg.vcycle(cycles)
    mg.sync()
    dt = (time.perf_counter() - t1) / cycles
    r1 = mg.rss()
    fac = (r1 / r0) ** (0.5 / cycles)
    import math
    need = math.ceil(math.log(1e-8) / math.log(fac)) if fac < 1 else float("inf")
    sizes = [mg.get_n_dofs(l) for l in range(L)]
    print(f"RS dim={dim} n={n} dofs={n**dim} levels={L} sizes={sizes[:4]}..{sizes[-1]} smoother={sm}: setup {setup:.2f}s, "
          f"{dt*1e3:.3f} ms/V-cycle = {1/dt:.1f} V-cycles/s, coarsest half-bw {mg.coarse_halfbw()}, "
          f"||r|| factor per cycle {fac:.4f}: 1e-8 in {need} cycles = {need*dt*1e3:.1f} ms", flush=True)
    if sm != "spgs":   # the V-cycle as M^-1 inside CG (reference README.md:127) from a zero guess
        mg.zero_vec(0, "u")
        mg.sync()
        t2 = time.perf_counter()
        _, it, rel = mg.pcg(1e-8, 200)
        t3 = time.perf_counter()
        print(f"   PCG on the same hierarchy: {it} iterations to relative residual {rel:.2e} in {(t3-t2)*1e3:.1f} ms "
              f"(incl. copying the solution back)", flush=True)
    mg.close()


if len(sys.argv) > 1 and sys.argv[1] == "rs":
    for n in (512, 1024, 2048):
        run_rs(n, "multicolor")
        run_rs(n, "jacobi")
    run_rs(1024, "spgs", 5)
    run_rs(64, "jacobi", 10, dim=3)
elif len(sys.argv) > 1 and sys.argv[1] == "all":
    run(2, 128, 3, "spgs", 20)                         # BASELINE config 1 (exact kernel: small)
    run(2, 1024, 6, "spgs", 5)                         # config 2, the reference's default smoother
    run(2, 1024, 6, "spgs", 5, exact_gs=True, exact_coarse_solve=True)   # ... parity mode
    run(2, 1024, 6, "jacobi1", 20)                     # config 2, true Jacobi 1+1
    run(2, 1024, 6, "jacobi1", 20, exact_coarse_solve=True)
    run(2, 1024, 12, "jacobi", 20)
    run(2, 1024, 6, "multicolor", 10)
    run(2, 4096, 16, "jacobi", 20)                     # config 3 (bench.py headline)
    run(2, 4096, 9, "jacobi", 20)
    run(2, 4096, 16, "multicolor", 10)
    run(2, 8192, 18, "jacobi", 10)                     # config 4 grid
    run(2, 8192, 9, "jacobi", 10)
    run(2, 8192, 18, "multicolor", 10)                 # config 4
    run(2, 8192, 9, "multicolor", 10)
    run(3, 256, 17, "jacobi", 10)
    run(3, 512, 20, "jacobi", 10)                      # config 5 grid on ONE GPU
else:
    dim, n, L, sm = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    run(dim, n, L, sm, int(sys.argv[5]) if len(sys.argv) > 5 else 10)
