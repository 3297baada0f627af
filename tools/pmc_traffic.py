#!/usr/bin/env python3
"""Builds profiles/<tag>_pmc_traffic.{md,json} from two rocprofv3 PMC passes
(--pmc FETCH_SIZE and --pmc WRITE_SIZE, each with --kernel-trace) of bench.py.
usage: pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <tag> [layout note]"""
import csv
import json
import re
import sys
from collections import defaultdict


def load(path):
    g = defaultdict(list)
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].split("(")[0].replace("void amg_hip::", "").replace("amg_hip::", "")
        g[(name, int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
    return g


def rows_of(name, grid):
    """rows one launch covers: K-Dict lanes own R rows (last template argument)."""
    m = re.match(r"dict_kernel<\d+, \d+, \d+, \w+, (\d+)>", name)
    return grid * int(m.group(1)) if m else grid


def format_bytes(name, rows):
    """bytes the layout itself must move per sweep: matrix stream + f + x + out"""
    m = re.match(r"dict_kernel<(\d+), (\d+),", name)
    if m:
        return rows * (8 * int(m.group(2)) + 24)
    return None


def main():
    f, w, tag = load(sys.argv[1]), load(sys.argv[2]), sys.argv[3]
    note = sys.argv[4] if len(sys.argv) > 4 else "K-Dict (dictionary-coded rows)"
    # 4096^2 hierarchy: rows and nnz of levels 0, 1, 2 (CSR formula 12 nnz + 28 n)
    lv = {16777216: 83869696, 8388607: 75485173, 4194303: 37742581}
    lv_pad = {16777216: 16777216, 8388608: 8388607, 4194304: 4194303}
    lines = [f"# rocprofv3 PMC traffic, {tag} (MI355X, bench.py --steps 4 --warmup 1, 4096^2, {note})", "",
             "Two separate passes (`rocprofv3 --pmc FETCH_SIZE --kernel-trace` and `--pmc WRITE_SIZE --kernel-trace`),",
             "values per dispatch in KB as rocprofv3 reports them.  On gfx950 FETCH_SIZE counts 128-B requests at 64 B",
             "(MI355X_MICROARCH.md, HBM section; calibration in r01_pmc_traffic.md): reads = 2 x FETCH_SIZE.  WRITE_SIZE is exact.", "",
             "| kernel | rows | calls | FETCH_SIZE raw MB | reads (x2) MB | WRITE_SIZE MB | traffic MB | format MB | algorithmic MB (CSR formula) | traffic / algorithmic |",
             "|---|---|---|---|---|---|---|---|---|---|"]
    res = {}
    for key in sorted(f, key=lambda k: -sum(f[k])):
        name, grid = key
        if not (name.startswith("sell_kernel") or name.startswith("dict_kernel")) or key not in w:
            continue
        rows = lv_pad.get(rows_of(name, grid))
        if rows is None:
            continue
        alg = 12.0 * lv[rows] + 28.0 * rows
        fr = sum(f[key]) / len(f[key]) * 1024
        wr = sum(w[key]) / len(w[key]) * 1024
        tr = 2 * fr + wr
        fb = format_bytes(name, rows)
        lines.append(f"| {name} | {rows} | {len(f[key])} | {fr / 1e6:.1f} | {2 * fr / 1e6:.1f} | {wr / 1e6:.1f} | "
                     f"{tr / 1e6:.1f} | {'%.1f' % (fb / 1e6) if fb else '-'} | {alg / 1e6:.1f} | {tr / alg:.3f} |")
        res[f"{name}@{rows_of(name, grid)}"] = {"fetch_raw_bytes": fr, "read_bytes_corrected": 2 * fr, "write_bytes": wr,
                                                "traffic_bytes": tr, "format_bytes": fb, "algorithmic_bytes": alg}
    lines += ["", "Template arguments: sell_kernel<mode, 16-bit columns, non-temporal>; dict_kernel<mode, code words per row,",
              "entries decoded per row, non-temporal, rows per lane>; mode 1 = Jacobi sweep, 0 = residual, 3 = rss terms.",
              "`format MB` = what the layout must move per sweep (codes + f + x + out); `algorithmic MB` = the CSR-formula",
              "figure (12 nnz + 28 n, SURVEY 8(d)) that `roofline.achieved` is quoted on.", ""]
    open(f"profiles/{tag}_pmc_traffic.md", "w").write("\n".join(lines))
    json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), FETCH_SIZE x2 gfx950 correction",
               "n": 4096, "layout": note, "kernels": res}, open(f"profiles/{tag}_pmc_traffic.json", "w"), indent=1)
    print("\n".join(lines))


if __name__ == "__main__":
    main()
