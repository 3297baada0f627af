#!/usr/bin/env python3
"""Builds profiles/<tag>_pmc_traffic.{md,json} from two rocprofv3 PMC passes
(--pmc FETCH_SIZE and --pmc WRITE_SIZE, each with --kernel-trace) of bench.py at 4096^2.
usage: pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <tag> [note]

The JSON carries `source_sha16` (hash of the device sources, bench.source_sha16()): bench.py
quotes a record as roofline.traffic only for the build it was measured on."""
import csv
import json
import os
import re
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def load(path):
    g = defaultdict(list)
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].split("(")[0].replace("void amg_hip::", "").replace("amg_hip::", "")
        g[(name, int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
    return g


N = 4096
LEVEL_ROWS = [16777216, 8388607, 4194303, 2097151, 1048575]          # 4096^2 hierarchy
LEVEL_NNZ = [83869696, 75485173, 37742581, 18871285, 9435637]
PITCH = [4096, 2048, 1024, 512, 256]


def patch_grid(l):
    """threads of a K-Patch launch on level l: tiles of 42 lines x 64 columns, 432 threads each"""
    n, m = LEVEL_ROWS[l], PITCH[l]
    lines = (n + m - 1) // m
    return ((lines + 41) // 42) * (m // 64) * 432


def classify(name, grid):
    """(kind, level, code words) of a launch, or (None, None, 0)"""
    m = re.match(r"patch_(down|up)_kernel<(\d+), ", name)
    if m:
        for l in range(len(LEVEL_ROWS)):
            if patch_grid(l) == grid:
                first = ", true, " in name and m.group(1) == "down"
                return ("patch_down_first" if first else "patch_" + m.group(1)), l, 0
        return None, None, 0
    rows, kind, words = None, None, 0
    m = re.match(r"dict_kernel<(\d+), (\d+), \d+, \w+, (\d+)>", name)
    if m:
        rows, words = grid * int(m.group(3)), int(m.group(2))
        kind = "resid" if m.group(1) == "0" else "sweep"
    m = re.match(r"dict_(resid_restrict|jacobi_prolong)_kernel<(\d+), \d+, \w+, (\d+)>", name)
    if m:   # overlapping tiles: 256 R rows each, stride 256 R - 2
        r = int(m.group(3))
        rows, kind, words = grid // 256 * (256 * r - 2), m.group(1), int(m.group(2))
    if name.startswith("sell_kernel"):
        rows, kind = grid, "sell"
    if rows is None:
        return None, None, 0
    for l, n in enumerate(LEVEL_ROWS):
        if abs(rows - n) <= 0.01 * n + 1024:
            return kind, l, words
    return None, None, 0


def must_move(kind, l):
    """bytes the launch has to move once (what bench.py's roofline.achieved is quoted on)"""
    n = LEVEL_ROWS[l]
    nH = LEVEL_ROWS[l + 1] if l + 1 < len(LEVEL_ROWS) else (n + 1) // 2 - 1
    if kind in ("sweep", "resid"):           # 1 B row type + f + x + out
        return n * 25
    if kind == "resid_restrict":             # r not stored; + f_H, first coarse sweep, coarse diagonal
        return n * 17 + nH * 24
    if kind == "jacobi_prolong":             # + read-modify-write of the finer u
        return n * 25 + LEVEL_ROWS[l - 1] * 16
    if kind in ("patch_down_first", "patch_down"):   # x, f, type, smoothed u; f_H, u_H (the coarse diagonal is a
        return n * 25 + nH * 16                      # kernel argument under interior tiles: > 99 % of them)
    if kind == "patch_up":                   # x, f, type, u_H in; u out
        return n * 25 + nH * 8
    if kind == "sell":
        return 12 * LEVEL_NNZ[l] + 28 * n
    return None


def main():
    from bench import source_sha16
    f, w, tag = load(sys.argv[1]), load(sys.argv[2]), sys.argv[3]
    note = sys.argv[4] if len(sys.argv) > 4 else "default layout (dictionary-coded rows, K-Patch on levels 0-3)"
    lines = [f"# rocprofv3 PMC traffic, {tag} (MI355X, bench.py, 4096^2, {note})", "",
             "Two separate passes (`rocprofv3 --pmc FETCH_SIZE --kernel-trace` and `--pmc WRITE_SIZE --kernel-trace`),",
             "values per dispatch in KB as rocprofv3 reports them.  On gfx950 FETCH_SIZE counts 128-B requests at 64 B",
             "(MI355X_MICROARCH.md, HBM section; calibration in r01_pmc_traffic.md): reads = 2 x FETCH_SIZE.  WRITE_SIZE is exact.", "",
             "| kernel | level | calls | FETCH_SIZE raw MB | reads (x2) MB | WRITE_SIZE MB | traffic MB | must move MB | traffic / must move | CSR formula of one sweep MB |",
             "|---|---|---|---|---|---|---|---|---|---|"]
    res = {}
    for key in sorted(f, key=lambda k: -sum(f[k])):
        name, grid = key
        kind, l, _ = classify(name, grid)
        if kind is None or key not in w:
            continue
        alg = 12.0 * LEVEL_NNZ[l] + 28.0 * LEVEL_ROWS[l]
        fr = sum(f[key]) / len(f[key]) * 1024
        wr = sum(w[key]) / len(w[key]) * 1024
        tr = 2 * fr + wr
        mm = must_move(kind, l)
        lines.append(f"| {name} | {l} | {len(f[key])} | {fr / 1e6:.1f} | {2 * fr / 1e6:.1f} | {wr / 1e6:.1f} | "
                     f"{tr / 1e6:.1f} | {mm / 1e6:.1f} | {tr / mm:.2f} | {alg / 1e6:.1f} |")
        res[f"{name}@L{l}"] = {"fetch_raw_bytes": fr, "read_bytes_corrected": 2 * fr, "write_bytes": wr,
                               "traffic_bytes": tr, "must_move_bytes": mm, "csr_formula_bytes": alg}
    lines += ["", "`must move` = what one launch has to read and write once (bench.py roofline.algorithmic_bytes_per_launch):",
              "dict_kernel: 1 B row type + f + x + out per row; the fused forms add their transfer operands;",
              "patch_down_kernel<slots, mask, first, nt>: the level's whole down-leg (x, f, type in; smoothed u, f_H, first coarse",
              "sweep out; the coarse diagonal is a kernel argument under interior tiles), patch_up_kernel: the up-leg (x, f, type,",
              "u_H in; u out).  traffic / must move",
              "above 1 = halo re-reads that miss the L2 / Infinity Cache.", ""]
    out_dir = os.environ.get("PMC_OUT_DIR", os.path.join(ROOT, "profiles"))
    os.makedirs(out_dir, exist_ok=True)
    open(os.path.join(out_dir, f"{tag}_pmc_traffic.md"), "w").write("\n".join(lines))
    json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), FETCH_SIZE x2 gfx950 correction",
               "n": N, "layout": note, "source_sha16": source_sha16(), "kernels": res},
              open(os.path.join(out_dir, f"{tag}_pmc_traffic.json"), "w"), indent=1)
    print("\n".join(lines))


if __name__ == "__main__":
    main()
