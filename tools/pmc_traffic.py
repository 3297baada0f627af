#!/usr/bin/env python3
"""Builds profiles/<tag>_pmc_traffic.{md,json} from two rocprofv3 PMC passes
(--pmc FETCH_SIZE and --pmc WRITE_SIZE, each with --kernel-trace) of bench.py.
usage: pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <tag> [calibration.csv]"""
import csv
import json
import sys
from collections import defaultdict


def load(path):
    g = defaultdict(list)
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].split("(")[0].replace("void amg_hip::", "").replace("amg_hip::", "")
        g[(name, int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
    return g


def main():
    f, w, tag = load(sys.argv[1]), load(sys.argv[2]), sys.argv[3]
    n0, nnz0 = 16777216, 83869696
    n1, nnz1 = 8388607, 75485173
    alg = {16777216: 12.0 * nnz0 + 28.0 * n0, 8388608: 12.0 * nnz1 + 28.0 * n1}
    lines = [f"# rocprofv3 PMC traffic, {tag} (MI355X, bench.py --steps 4 --warmup 1, 4096^2, SELL-64, 16-bit relative columns)", "",
             "Two separate passes (`rocprofv3 --pmc FETCH_SIZE --kernel-trace` and `--pmc WRITE_SIZE --kernel-trace`),",
             "values per dispatch in KB as rocprofv3 reports them.  On gfx950 FETCH_SIZE counts 128-B requests at 64 B",
             "(MI355X_MICROARCH.md, HBM section; calibration below): reads = 2 x FETCH_SIZE.  WRITE_SIZE is exact.", "",
             "| kernel | grid | calls | FETCH_SIZE raw MB | reads (x2) MB | WRITE_SIZE MB | traffic MB | algorithmic MB (CSR formula) | ratio |",
             "|---|---|---|---|---|---|---|---|---|"]
    res = {}
    for key in sorted(f, key=lambda k: -sum(f[k])):
        name, grid = key
        if not name.startswith("sell_kernel") or grid not in alg or key not in w:
            continue
        fr = sum(f[key]) / len(f[key]) * 1024
        wr = sum(w[key]) / len(w[key]) * 1024
        tr = 2 * fr + wr
        lines.append(f"| {name} | {grid} | {len(f[key])} | {fr / 1e6:.1f} | {2 * fr / 1e6:.1f} | {wr / 1e6:.1f} | "
                     f"{tr / 1e6:.1f} | {alg[grid] / 1e6:.1f} | {tr / alg[grid]:.3f} |")
        res[f"{name}@{grid}"] = {"fetch_raw_bytes": fr, "read_bytes_corrected": 2 * fr, "write_bytes": wr,
                                 "traffic_bytes": tr, "algorithmic_bytes": alg[grid]}
    lines += ["", "Mode <1,...> = Jacobi sweep, <0,...> = residual.  The SELL layout reads no row pointer and, with 16-bit",
              "relative column indices, 10 instead of 12 bytes per entry, so the measured traffic sits below the",
              "CSR-formula figure that `roofline.achieved` is quoted on; level 1 additionally drops its exact-zero entries.", ""]
    open(f"profiles/{tag}_pmc_traffic.md", "w").write("\n".join(lines))
    json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), FETCH_SIZE x2 gfx950 correction",
               "n": 4096, "layout": "sell64-idx16", "kernels": res}, open(f"profiles/{tag}_pmc_traffic.json", "w"), indent=1)
    print("\n".join(lines))


if __name__ == "__main__":
    main()
