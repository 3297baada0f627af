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


def n_levels_for(n, coarsest_max=511):
    """Smallest level count whose coarsest level has <= coarsest_max dofs
    (n_H = (n_h+1)/2 - 1, multigrid.hpp:127-130)."""
    dofs, levels = n * n, 1
    while dofs > coarsest_max:
        dofs = (dofs + 1) // 2 - 1
        levels += 1
    return levels


def cpu_baseline(args):
    """The CPU oracle (Eigen-order restatement of the reference, 1 thread -- the
    reference is serial) timed on a bounded sample of the same workload."""
    from oracle import oracle as O
    n = args.cpu_n
    L = n_levels_for(n)
    A, b = O.laplacian(n), O.rhs(n)
    mg = O.Multigrid(A, b, L, smoother=O.SM_TRUE_JACOBI, smoother_iters=args.sweeps,
                     omega=args.omega)
    mg.time_vcycles(1)  # warm-up
    reps = args.cpu_cycles
    times = sorted(mg.time_vcycles(1) for _ in range(reps))
    med = times[len(times) // 2]
    scale = (n * n) / float(args.n * args.n)   # V-cycle cost is linear in the dofs
    return {
        "value": (1.0 / med) * scale,
        "unit": "V-cycles/s",
        "cores": 1,
        "host_cores": os.cpu_count(),
        "kind": "port",
        "sample": (f"{reps} vcycle() calls (median) of the CPU oracle on the {n}x{n} instance of "
                   f"the same workload ({L} levels, same smoother), rate scaled by "
                   f"{n * n}/{args.n * args.n} dofs to the {args.n}x{args.n} problem; "
                   f"measured {1.0 / med:.3f} V-cycles/s at {n}x{n}"),
    }


def run_single(args):
    import amg_ctypes as amg
    if amg.device_count() < 1:
        raise SystemExit("bench.py needs a HIP device (libamg_hip.so has no CPU fallback)")
    if args.no_nt:
        amg.lib().amg_hip_set_nontemporal(0)
    amg.set_dict_rows(args.dict_rows)
    amg.set_row_types(not args.no_row_types)
    amg.set_xcd_mapping(not args.no_xcd_map)
    amg.set_default_layout({"auto": amg.LAYOUT_AUTO, "csr": amg.LAYOUT_CSR, "sell": amg.LAYOUT_SELL,
                            "dict": amg.LAYOUT_DICT}[args.layout])
    t0 = time.time()
    colptr, rowind, val = amg.laplacian(args.n)
    b = amg.rhs(args.n)
    L = args.levels or n_levels_for(args.n)
    if args.smoother == "multicolor":   # BASELINE configs[3] smoother; one symmetric colour pass
        mg = amg.Multigrid(colptr, rowind, val, b, L, smoother=amg.SM_MULTICOLOR_GS,
                           smoother_iters=1, use_graph=not args.no_graph)
    else:
        mg = amg.Multigrid(colptr, rowind, val, b, L, smoother=amg.SM_JACOBI,
                           smoother_iters=args.sweeps, omega=args.omega, use_graph=not args.no_graph,
                           fast_coarse_solve=args.fast_coarse)
    setup_s = time.time() - t0
    del colptr, rowind, val
    mg.vcycle(args.warmup)
    mg.sync()
    rss0 = mg.rss()   # after the warm-up cycles (the first cycle from u=0 raises rss)
    t1 = time.perf_counter()
    mg.vcycle(args.steps)
    mg.sync()
    dt = time.perf_counter() - t1
    # dominant kernel: level-0 Jacobi sweep, HIP events on the solver's stream
    if args.smoother == "jacobi":
        avg_ms, min_ms = mg.profile_fine_sweep(args.profile_launches)
    else:
        avg_ms, min_ms = float("nan"), float("nan")
    cyc_bytes, sweep_bytes = mg.cycle_bytes()
    rss = mg.rss()
    if args.warmup >= 1 and args.smoother == "jacobi" and not (rss < rss0):
        raise SystemExit(f"V-cycle iteration is not converging (rss {rss0:.3e} -> {rss:.3e})")
    sizes = [mg.get_n_dofs(l) for l in range(L)]
    achieved = sweep_bytes / (avg_ms * 1e-3) / 1e9
    lay, mat_bytes = mg.level_layout(0)
    lay_name = {amg.LAYOUT_CSR: "csr", amg.LAYOUT_SELL: "sell", amg.LAYOUT_DICT: "dict"}[lay]
    # what the format has to move per sweep: matrix stream + f + x + out (8 B each per row)
    format_bytes = mat_bytes + 24 * sizes[0]
    kernel = {"dict": ("dict_kernel<CSR_JACOBI, 1 code word, 5 entries, nt, 2 rows/lane> (level-0 Jacobi sweep, "
                       "dictionary-coded rows, one byte per row)", "r01h_pmc_traffic", "dict_kernel<1, 1, 5, true, 2>@L0"),
              "sell": ("sell_kernel<CSR_JACOBI, idx16, nt> (level-0 Jacobi sweep, SELL-64 panels)",
                       "r01c_pmc_traffic", "sell_kernel<1, true, true>@16777216"),
              "csr": ("csr_stage_kernel<CSR_JACOBI> (level-0 Jacobi sweep, LDS-staged CSR)", None, None)}[lay_name]
    # HBM traffic of that kernel from the committed PMC passes (rocprofv3 cannot run
    # inside this process); only quoted for the configuration it was measured on.
    traffic, traffic_src = None, None
    pmc = os.path.join(ROOT, "profiles", f"{kernel[1]}.json")
    if kernel[1] and os.path.exists(pmc) and args.n == 4096 and not args.no_nt:
        rec = json.load(open(pmc))
        k = rec["kernels"].get(kernel[2])
        if k and rec.get("n") == args.n:
            traffic, traffic_src = k["traffic_bytes"], f"profiles/{kernel[1]}.md: " + rec["source"]
    out = {
        "metric": "V-cycles/sec, 2D Poisson N=4096^2 (fine-grid smoother HBM GB/s under roofline)",
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
            "workload": (f"2D 5-point Poisson {args.n}x{args.n} (Grid::laplacian/rhs), true Jacobi "
                         f"smoother omega={args.omega} {args.sweeps}+{args.sweeps} sweeps, "
                         f"{L}-level V-cycle, coarsest {sizes[-1]} dofs, fp64, 1xMI355X"),
            "n": args.n, "levels": L, "smoother": args.smoother, "omega": args.omega,
            "sweeps": args.sweeps, "graph": not args.no_graph, "fast_coarse_solve": args.fast_coarse,
            "cycle_algorithmic_bytes": cyc_bytes, "cycle_GBps": cyc_bytes / (dt / args.steps) / 1e9,
            "setup_seconds": setup_s, "rss_after_warmup": rss0, "rss_after_steps": rss,
        },
        "roofline": {
            "bound": "hbm",
            "kernel": kernel[0],
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic,
            "traffic_source": traffic_src,
            "algorithmic_bytes_per_launch": sweep_bytes,
            "layout": lay_name,
            "format_bytes_per_launch": format_bytes,
            "format_GBps": format_bytes / (avg_ms * 1e-3) / 1e9,
            "hbm_GBps_measured": (traffic / (avg_ms * 1e-3) / 1e9) if traffic else None,
            "hbm_frac_measured": (traffic / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
            "note": ("achieved/frac use the CSR-formula bytes of SURVEY 8(d) (12 nnz + 28 n per sweep), i.e. "
                     "the CSR-equivalent rate; the dictionary-coded layout moves format_bytes_per_launch "
                     "(1 B of row type + f + x + out per row), so frac can exceed 1 -- hbm_*_measured is the "
                     "PMC traffic over the same launch time against the 8 TB/s peak") if lay_name == "dict" else None,
            "avg_launch_ms": avg_ms,
            "min_launch_ms": min_ms,
            "launches_timed": args.profile_launches,
        },
    }
    mg.close()
    # the same workload with the level matrices as plain CSR panels (SELL-64, the layout the
    # CSR-formula bytes describe): a second, shorter measurement in the same run
    if lay_name == "dict" and args.smoother == "jacobi" and not args.no_csr_ref:
        colptr, rowind, val = amg.laplacian(args.n)
        ref = amg.Multigrid(colptr, rowind, val, b, L, smoother=amg.SM_JACOBI, smoother_iters=args.sweeps,
                            omega=args.omega, use_graph=not args.no_graph, layout=amg.LAYOUT_SELL)
        del colptr, rowind, val
        ref.vcycle(args.warmup)
        ref.sync()
        t2 = time.perf_counter()
        k = max(10, args.steps // 2)
        ref.vcycle(k)
        ref.sync()
        dt2 = time.perf_counter() - t2
        ref_ms, _ = ref.profile_fine_sweep(max(8, args.profile_launches // 2))
        rss_ref = ref.rss()
        ref.close()
        out["config"]["csr_layout_reference"] = {
            "layout": "sell (CSR sliced into 64-row panels, 16-bit relative columns)",
            "vcycles_per_sec": k / dt2, "ms_per_step": dt2 / k * 1e3, "steps": k,
            "fine_sweep_ms": ref_ms, "fine_sweep_GBps": sweep_bytes / (ref_ms * 1e-3) / 1e9,
            "fine_sweep_frac_of_peak": sweep_bytes / (ref_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "rss_after_warmup_plus_steps": rss_ref,
        }
    if not args.no_cpu:
        out["cpu_baseline"] = cpu_baseline(args)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--grid", dest="n", type=int, default=4096, help="grid points per direction")
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
    ap.add_argument("--fast-coarse", action="store_true",
                    help="partitioned (parallel) coarse solve; then fewer levels pay off (--levels 13)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-csr-ref", action="store_true",
                    help="skip the second measurement with the plain-CSR (SELL-64) layout")
    ap.add_argument("--cpu-n", type=int, default=2048, help="grid of the CPU baseline sample")
    ap.add_argument("--cpu-cycles", type=int, default=5)
    ap.add_argument("--profile-launches", type=int, default=40)
    ap.add_argument("--comm", choices=["auto", "p2p", "ipc", "graph"], default="auto",
                    help="multi-GPU halo exchange: p2p = torch.distributed isend/irecv (RCCL); ipc = "
                         "direct hipIpc pushes with stream memory ops; graph = pushes + flags as "
                         "kernels, one hipGraph per rank; auto = time all, keep the fastest that "
                         "reproduces the p2p result bit for bit")
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
