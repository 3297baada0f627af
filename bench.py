#!/usr/bin/env python3
"""bench.py -- V-cycles/s of the MI355X-native multigrid V-cycle.

Workload (BASELINE.json metric / configs[2]): 2-D 5-point Poisson 4096 x 4096
(A = Grid::laplacian, b = Grid::rhs, u0 = 0), true (two-buffer) Jacobi smoother,
2 sweeps pre + 2 post on every level, 16-level hierarchy (coarsest 511 dofs),
fp64.  One "step" = one vcycle() (multigrid.hpp:263-305), rss excluded.

    python bench.py --gpus N --steps K --warmup W

N = 1: one process, one GPU.  N > 1: launched by torch.distributed.run, one rank
per GPU, row-block sharded hierarchy with RCCL halo exchange (see
algebraic-multigrid_amd/dist_vcycle.py); fixed global problem => strong scaling.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "algebraic-multigrid_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
CPU_FLAGS = ("-O2 -ffp-contract=off", "-O3 -march=native -ffp-contract=off")


def metric_string(n, dim=2):
    if dim == 3:
        return f"V-cycles/sec, 3D 7-point Poisson N={n}^3 (fine-grid smoother HBM GB/s under roofline)"
    return f"V-cycles/sec, 2D Poisson N={n}^2 (fine-grid smoother HBM GB/s under roofline)"


def source_sha16():
    """Identifies the kernel build a PMC record belongs to: hash of the device sources."""
    import hashlib
    h = hashlib.sha256()
    for f in ("kernels.hip", "kernels.hpp", "solver.cpp", "host_setup.cpp", "host_setup.hpp"):
        with open(os.path.join(PKG, "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def pmc_traffic(kernel_prefix, n):
    """HBM bytes per launch of the kernel whose name starts with `kernel_prefix`, from a
    committed rocprofv3 PMC record (profiles/*_pmc_traffic.json; rocprofv3 cannot run inside
    this process).  A record is only quoted for the build and the grid it was measured on:
    its source_sha16 must equal the hash of the current device sources."""
    import glob
    sha = source_sha16()
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")), reverse=True):
        try:
            rec = json.load(open(path))
        except Exception:
            continue
        if rec.get("source_sha16") != sha or rec.get("n") != n:
            continue
        for key, k in rec.get("kernels", {}).items():
            if key.startswith(kernel_prefix) and key.endswith("@L0"):
                return k["traffic_bytes"], (f"profiles/{os.path.basename(path)[:-5]}.md [{key}]: " + rec["source"])
    return None, f"no PMC record for this build (source_sha16 {sha}); see tools/pmc_traffic.py"


def n_levels_for(n, coarsest_max=511, dim=2):
    """Smallest level count whose coarsest level has <= coarsest_max dofs
    (n_H = (n_h+1)/2 - 1, multigrid.hpp:127-130)."""
    dofs, levels = n ** dim, 1
    while dofs > coarsest_max:
        dofs = (dofs + 1) // 2 - 1
        levels += 1
    return levels


def cpu_baseline(args):
    """The CPU oracle (Eigen-order restatement of the reference, 1 thread -- the reference
    is serial) timed on the benchmarked instance itself (--grid; SURVEY 8(d): vcycle() only,
    1 warm-up, median), once per flag set of CPU_FLAGS.  `value` is the faster of the two."""
    from oracle import oracle as O
    n = args.cpu_n or args.n
    dim = args.dim
    L = args.levels or n_levels_for(n, dim=dim)
    A, b = O.laplacian(n, dim), O.rhs(n, dim)
    colors = None
    if args.smoother == "multicolor":   # the twin replays the product's greedy colouring (host code)
        import amg_ctypes as amg
        h = amg.Multigrid(A.colptr, A.rowind, A.val, b, L, smoother=amg.SM_MULTICOLOR_GS, host_only=True)
        colors = [h.get_colors(l) for l in range(L)]
        h.close()
    runs = {}
    for flags in CPU_FLAGS:
        lib = O.lib() if flags == CPU_FLAGS[0] else O.lib_variant(flags)
        t0 = time.time()
        if colors is None:
            mg = O.Multigrid(A, b, L, smoother=O.SM_TRUE_JACOBI, smoother_iters=args.sweeps,
                             omega=args.omega, library=lib)
        else:
            mg = O.Multigrid(A, b, L, smoother=O.SM_MULTICOLOR, smoother_iters=1, library=lib)
            for l, (c, nc) in enumerate(colors):
                mg.set_colors(l, c, nc)
        setup = time.time() - t0
        mg.time_vcycles(1)  # warm-up
        times = sorted(mg.time_vcycles(1) for _ in range(args.cpu_cycles))
        del mg
        runs[flags] = {"vcycles_per_sec": 1.0 / times[len(times) // 2], "setup_seconds": setup}
    best = max(runs, key=lambda k: runs[k]["vcycles_per_sec"])
    scale = (n ** dim) / float(args.n ** dim)   # 1 unless --cpu-n asks for a smaller sample
    return {
        "value": runs[best]["vcycles_per_sec"] * scale,
        "unit": "V-cycles/s",
        "cores": 1,
        "host_cores": os.cpu_count(),
        "kind": "port",
        "flags": best,
        "by_flags": runs,
        "sample": (f"{args.cpu_cycles} vcycle() calls (median, after 1 warm-up) of the CPU oracle "
                   f"(g++ {best}) on the {n}^{dim} instance ({L} levels, same smoother, same "
                   f"omega)" + ("" if n == args.n else
                                f", rate scaled by {n ** dim}/{args.n ** dim} dofs to {args.n}^{dim}")),
    }


def run_single(args):
    import amg_ctypes as amg
    if amg.device_count() < 1:
        raise SystemExit("bench.py needs a HIP device (libamg_hip.so has no CPU fallback)")
    if args.no_nt:
        amg.lib().amg_hip_set_nontemporal(0)
    amg.set_dict_rows(args.dict_rows)
    if args.patch_min_rows is not None:
        amg.set_patch_min_rows(args.patch_min_rows)
    amg.set_row_types(not args.no_row_types)
    amg.set_xcd_mapping(not args.no_xcd_map)
    amg.set_default_layout({"auto": amg.LAYOUT_AUTO, "csr": amg.LAYOUT_CSR, "sell": amg.LAYOUT_SELL,
                            "dict": amg.LAYOUT_DICT}[args.layout])
    # setup on the device end to end (generator, Galerkin chain, encoder: amg_hip_create_poisson);
    # --host-setup hands Grid-generated host arrays to the general constructor instead
    t0 = time.time()
    dim = args.dim
    L = args.levels or n_levels_for(args.n, dim=dim)
    kw = (dict(smoother=amg.SM_MULTICOLOR_GS, smoother_iters=1) if args.smoother == "multicolor" else
          dict(smoother=amg.SM_JACOBI, smoother_iters=args.sweeps, omega=args.omega,
               fast_coarse_solve=args.fast_coarse))
    if args.host_setup:
        colptr, rowind, val = amg.laplacian(args.n, dim)
        b = amg.rhs(args.n, dim)
        mg = amg.Multigrid(colptr, rowind, val, b, L, use_graph=not args.no_graph, **kw)
        del colptr, rowind, val
    else:
        mg = amg.Multigrid.poisson(args.n, L, dim=dim, use_graph=not args.no_graph, **kw)
    mg.sync()
    setup_s = time.time() - t0
    mg.vcycle(args.warmup)
    mg.sync()
    rss0 = mg.rss()   # after the warm-up cycles (the first cycle from u=0 raises rss)
    t1 = time.perf_counter()
    mg.vcycle(args.steps)
    mg.sync()
    dt = time.perf_counter() - t1
    cyc_bytes, sweep_bytes = mg.cycle_bytes()
    rss = mg.rss()
    # every configuration must reduce rss over the timed cycles; the ratio is recorded
    if args.warmup >= 1 and not (rss < rss0):
        raise SystemExit(f"V-cycle iteration is not converging (rss {rss0:.3e} -> {rss:.3e})")
    sizes = [mg.get_n_dofs(l) for l in range(L)]
    lay, mat_bytes = mg.level_layout(0)
    lay_name = {amg.LAYOUT_CSR: "csr", amg.LAYOUT_SELL: "sell", amg.LAYOUT_DICT: "dict"}[lay]
    must = mg.cycle_must_move()
    problem = (f"2D 5-point Poisson {args.n}x{args.n} (Grid::laplacian/rhs)" if dim == 2 else
               f"3D 7-point Poisson {args.n}^3 (the 7-point analogue of Grid::laplacian/rhs)")
    out = {
        "metric": metric_string(args.n, dim),
        "value": args.steps / dt,
        "unit": "V-cycles/s",
        "n_gpus": 1,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": (problem + ", "
                         + (f"true Jacobi smoother omega={args.omega} {args.sweeps}+{args.sweeps} sweeps"
                            if args.smoother == "jacobi" else "multicolour symmetric GS 1+1 passes")
                         + f", {L}-level V-cycle, coarsest {sizes[-1]} dofs, fp64, 1xMI355X; "
                           "smoother has no counterpart in the reference: pinned to the oracle twin"),
            "n": args.n, "dim": dim, "levels": L, "smoother": args.smoother, "omega": args.omega,
            "sweeps": args.sweeps, "graph": not args.no_graph, "coarse_solve": mg.coarse_solve_kind(),
            "layout": lay_name, "setup_seconds": setup_s,
            "rss_after_warmup": rss0, "rss_after_steps": rss,
            "rss_ratio_per_cycle": (rss / rss0) ** (1.0 / args.steps) if rss0 > 0 else None,
            "csr_formula_cycle_bytes": cyc_bytes,
            # the whole cycle against the HBM roof: bytes every launch of the cycle has to move in the
            # layout it streams (amg_hip_cycle_must_move) / measured time per cycle
            "whole_cycle": {"must_move_bytes": must, "GBps": must / (dt / args.steps) / 1e9,
                            "frac_of_hbm_peak": must / (dt / args.steps) / 1e9 / HBM_PEAK_GBS},
        },
        "roofline": fine_sweep_roofline(amg, mg, args, lay_name, mat_bytes, sizes[0], sweep_bytes),
    }
    mg.close()
    # the same workload with the level matrices as plain CSR panels (SELL-64): the layout
    # SURVEY 8(d)'s CSR-formula bytes describe; a second, shorter measurement in the same run
    if lay_name == "dict" and args.smoother == "jacobi" and not args.no_csr_ref:
        ref = amg.Multigrid.poisson(args.n, L, dim=dim, smoother=amg.SM_JACOBI, smoother_iters=args.sweeps,
                                    omega=args.omega, use_graph=not args.no_graph, layout=amg.LAYOUT_SELL)
        ref.vcycle(args.warmup)
        ref.sync()
        t2 = time.perf_counter()
        k = max(10, args.steps // 2)
        ref.vcycle(k)
        ref.sync()
        dt2 = time.perf_counter() - t2
        lay2, mat2 = ref.level_layout(0)
        roof2 = fine_sweep_roofline(amg, ref, args, "sell", mat2, sizes[0], sweep_bytes,
                                    launches=max(8, args.profile_launches // 2))
        rss_ref = ref.rss()
        must2 = ref.cycle_must_move()
        ref.close()
        out["config"]["csr_layout_reference"] = {
            "layout": "sell (CSR sliced into 64-row panels, 16-bit relative columns)",
            "vcycles_per_sec": k / dt2, "ms_per_step": dt2 / k * 1e3, "steps": k,
            "rss_after_warmup_plus_steps": rss_ref, "roofline": roof2,
            "whole_cycle": {"must_move_bytes": must2, "GBps": must2 / (dt2 / k) / 1e9,
                            "frac_of_hbm_peak": must2 / (dt2 / k) / 1e9 / HBM_PEAK_GBS},
        }
    if not args.no_cpu:
        out["cpu_baseline"] = cpu_baseline(args)
    return out


def fine_sweep_roofline(amg, mg, args, lay_name, mat_bytes, n0, csr_formula_bytes, launches=None):
    """`roofline` object of the dominant kernel: the level-0 Jacobi sweep(s).

    achieved = bytes ONE LAUNCH has to move / its average duration (HIP events on the solver's
    stream, amg_hip_profile_fine_sweep).  The bytes follow from what the kernel reads and
    writes, never from a layout it does not stream:
      sell / csr: SURVEY 8(d)'s CSR formula, 12 nnz + 28 n per sweep;
      dict:       matrix stream (1 B row type per row) + f + x + out (8 B each per row);
      K-Patch (temporal blocking: a level's down-leg in one launch) still has to move
      x, f and the smoothed u once plus the coarse vectors it produces -- the sweeps it
      saves are traffic it no longer causes; they show up in V-cycles/s, not in this
      fraction.  The library reports the figure (amg_hip_fine_sweep_info)."""
    avg_ms, min_ms, sweeps_per_launch, kname, must_move = mg.profile_fine_sweep(launches or args.profile_launches)
    if kname.startswith("patch_rb"):
        model = ("K-Patch form of the multicolour pass on the red-black level 0: one launch = the first colour "
                 "stage(s) of the level's down-leg: n*(1 B row type + x + f + out)")
    elif kname.startswith("march_kernel"):
        model = ("K-March: both Jacobi sweeps of the 3-D 7-point level 0 in one plane-marching launch: "
                 "n*(1 B row type + x + f + out)")
    elif kname.startswith("dict_gs_color") or kname.startswith("sell_kernel<5"):
        model = "rows of colour 0: matrix stream of those rows (code words / panels + dof id) + f + u written"
    elif kname.startswith("patch_down"):
        model = ("K-Patch down-leg of level 0 in one launch (2 Jacobi sweeps + residual + restriction + "
                 "first coarse sweep): n*(1 B row type + x + f + smoothed u) + n_H*(f_H + u_H + coarse diagonal)")
    elif lay_name == "dict":
        model = "n*(1 B row type + f + x + out) + tables"
    else:
        model = "12 nnz + 28 n (SURVEY 8(d) CSR formula)"
    achieved = must_move / (avg_ms * 1e-3) / 1e9
    traffic, traffic_src = (None, "not quoted: non-default kernel switches")
    if not args.no_nt:
        traffic, traffic_src = pmc_traffic(kname.split("<")[0], args.n)
    return {
        "bound": "hbm",
        "kernel": kname,
        "achieved": achieved,
        "peak": HBM_PEAK_GBS,
        "unit": "GB/s",
        "frac": achieved / HBM_PEAK_GBS,
        "traffic": traffic,
        "traffic_source": traffic_src,
        "algorithmic_bytes_per_launch": must_move,
        "bytes_model": model,
        "sweeps_per_launch": sweeps_per_launch,
        "layout": lay_name,
        "csr_formula_bytes_per_sweep": csr_formula_bytes,
        "csr_equivalent_GBps": csr_formula_bytes * sweeps_per_launch / (avg_ms * 1e-3) / 1e9,
        "hbm_GBps_measured": (traffic / (avg_ms * 1e-3) / 1e9) if traffic else None,
        "avg_launch_ms": avg_ms,
        "min_launch_ms": min_ms,
        "launches_timed": launches or args.profile_launches,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--grid", dest="n", type=int, default=4096, help="grid points per direction")
    ap.add_argument("--dim", type=int, default=2, choices=[2, 3],
                    help="2: Grid::laplacian (5-point); 3: the 7-point analogue (BASELINE config 5)")
    ap.add_argument("--levels", type=int, default=0, help="0 = coarsest <= 511 dofs")
    ap.add_argument("--omega", type=float, default=0.6,
                    help="Jacobi relaxation; must stay below 2/lambda_max(D^-1 A) ~ 0.67 on the "
                         "reference's flat-index Galerkin levels (DESIGN.md)")
    ap.add_argument("--sweeps", type=int, default=2, help="Jacobi sweeps per smooth() call")
    ap.add_argument("--smoother", choices=["jacobi", "multicolor"], default="jacobi")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--layout", choices=["auto", "csr", "sell", "dict"], default="auto",
                    help="device layout of the level matrices (auto: dictionary-coded rows where a "
                         "matrix qualifies, else SELL-64)")
    ap.add_argument("--no-nt", action="store_true", help="disable the non-temporal matrix stream")
    ap.add_argument("--no-xcd-map", action="store_true", help="K-Dict: plain blockIdx -> tile mapping")
    ap.add_argument("--no-row-types", action="store_true", help="K-Dict: first-level coding only")
    ap.add_argument("--dict-rows", type=int, default=2, choices=[1, 2], help="K-Dict rows per lane")
    ap.add_argument("--patch-min-rows", type=int, default=None,
                    help="K-Patch (temporal blocking) on levels of at least this many rows "
                         "(default 10^6; -1 = off)")
    ap.add_argument("--fast-coarse", action="store_true",
                    help="partitioned (parallel) coarse solve; then fewer levels pay off (--levels 13)")
    ap.add_argument("--host-setup", action="store_true",
                    help="build the hierarchy from host arrays (amg_hip_create) instead of on the device")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-csr-ref", action="store_true",
                    help="skip the second measurement with the plain-CSR (SELL-64) layout")
    ap.add_argument("--cpu-n", type=int, default=0,
                    help="grid of the CPU baseline sample (0 = the benchmarked --grid itself)")
    ap.add_argument("--cpu-cycles", type=int, default=3)
    ap.add_argument("--profile-launches", type=int, default=40)
    ap.add_argument("--window-levels", type=int, default=-1,
                    help="window sharding: number of distributed levels (-1 = every level of at least "
                         "--window-min-rows rows whose halo overhead stays below 35 %% of a block)")
    ap.add_argument("--window-min-rows", type=int, default=500000)
    ap.add_argument("--comm", choices=["safe", "auto", "p2p", "slab", "window", "library", "ipc", "graph"], default="safe",
                    help="multi-GPU halo exchange.  safe (default) = the configurations built on "
                         "plain RCCL calls only: replicated (nothing distributed: timed first, it is "
                         "the reference every other one must reproduce), p2p (a send/recv before every "
                         "sweep, residual and transfer) and slab (K-Patch levels over each rank's grid "
                         "lines + redundant halo: one grouped send/recv and one all-gather per cycle) and "
                         "window (every rank sets up and stores only its window of the distributed levels; "
                         "the only sharded form of --smoother multicolor and --dim 3); "
                         "the fastest one is reported.  library = safe + the slab / window cycles as ONE "
                         "C call each over the library's own RCCL communicator (amg_hip_slab_cycle, "
                         "amg_hip_window_cycle; run with a world of 1 only so far).  auto = additionally "
                         "ipc (hipIpc pushes + stream memory ops) and graph (pushes + flags as "
                         "kernels, one hipGraph per rank): experimental, never run on real xGMI "
                         "links; bounded by a watchdog that exits with status 3 when one hangs.  "
                         "Every candidate must reproduce the p2p result bit for bit")
    ap.add_argument("--slab-levels", type=int, default=-1,
                    help="slab sharding: number of K-Patch levels cut into row blocks (-1 = all of them)")
    ap.add_argument("--slab-patch-min-rows", type=int, default=0,
                    help="slab sharding: K-Patch threshold of the sharded solver (0 = every level whose "
                         "lines are at least 128 entries long: each rank only runs 1/N of a slab level; "
                         "compute-only estimates in tools/slab_estimate.py)")
    ap.add_argument("--dist-min-rows-ipc", type=int, default=5000000)
    ap.add_argument("--dist-min-rows-graph", type=int, default=250000)
    ap.add_argument("--comm-timeout", type=float, default=180.0,
                    help="seconds after which a hung alternative exchange mode is abandoned")
    ap.add_argument("--dist-min-rows", type=int, default=10000000,
                    help="multi-GPU: levels with fewer rows (in total) run redundantly on every rank")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 or world > 1:
        import dist_vcycle
        out = dist_vcycle.bench(args)
        if out is not None:
            print(json.dumps(out), flush=True)
        return
    out = run_single(args)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
