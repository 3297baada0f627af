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


LEVEL_ROWS = [16777216, 8388607, 4194303, 2097151, 1048575]   # 4096^2 hierarchy


def parse(name, grid):
    """(kind, rows the launch covers, code words per row) of a K-Dict launch."""
    m = re.match(r"dict_kernel<(\d+), (\d+), \d+, \w+, (\d+)>", name)
    if m:
        rows = grid * int(m.group(3))
        return ("resid" if m.group(1) == "0" else "sweep"), rows, int(m.group(2))
    m = re.match(r"dict_(resid_restrict|jacobi_prolong)_kernel<(\d+), \d+, \w+, (\d+)>", name)
    if m:   # overlapping tiles: 256 R rows each, stride 256 R - 2
        r = int(m.group(3))
        rows = grid // 256 * (256 * r - 2)
        return m.group(1), rows, int(m.group(2))
    if name.startswith("sell_kernel"):
        return "sell", grid, 0
    return None, grid, 0


def level_of(rows):
    for l, n in enumerate(LEVEL_ROWS):
        if abs(rows - n) <= 0.01 * n + 1024:
            return l
    return None


ROW_TYPES = True   # second-level coding: one byte per row instead of 8 * words


def format_bytes(kind, l, words):
    """bytes the layout itself must move per launch"""
    n = LEVEL_ROWS[l]
    mat = 1 if ROW_TYPES else 8 * words
    if kind in ("sweep", "resid"):          # row types / codes + f + x + out
        return n * (mat + 24)
    if kind == "resid_restrict":            # r is not stored (opt.keep_residual = 0);
        return n * (mat + 16) + LEVEL_ROWS[l + 1] * 24   # + f_H, first coarse sweep out, coarse diagonal in
    if kind == "jacobi_prolong":            # + read-modify-write of the finer u
        return n * (mat + 24) + LEVEL_ROWS[l - 1] * 16
    return None


def main():
    f, w, tag = load(sys.argv[1]), load(sys.argv[2]), sys.argv[3]
    note = sys.argv[4] if len(sys.argv) > 4 else "K-Dict (dictionary-coded rows, one byte per row)"
    # 4096^2 hierarchy: rows and nnz of levels 0, 1, 2 (CSR formula 12 nnz + 28 n)
    lv = {16777216: 83869696, 8388607: 75485173, 4194303: 37742581}
    lv_pad = {16777216: 16777216, 8388608: 8388607, 4194304: 4194303}
    lines = [f"# rocprofv3 PMC traffic, {tag} (MI355X, bench.py --steps 4 --warmup 1, 4096^2, {note})", "",
             "Two separate passes (`rocprofv3 --pmc FETCH_SIZE --kernel-trace` and `--pmc WRITE_SIZE --kernel-trace`),",
             "values per dispatch in KB as rocprofv3 reports them.  On gfx950 FETCH_SIZE counts 128-B requests at 64 B",
             "(MI355X_MICROARCH.md, HBM section; calibration in r01_pmc_traffic.md): reads = 2 x FETCH_SIZE.  WRITE_SIZE is exact.", "",
             "| kernel | level | calls | FETCH_SIZE raw MB | reads (x2) MB | WRITE_SIZE MB | traffic MB | format MB | algorithmic MB (CSR formula) | traffic / algorithmic |",
             "|---|---|---|---|---|---|---|---|---|---|"]
    res = {}
    nnz = {0: 83869696, 1: 75485173, 2: 37742581}
    for key in sorted(f, key=lambda k: -sum(f[k])):
        name, grid = key
        kind, rows, words = parse(name, grid)
        l = level_of(rows) if kind else None
        if l is None or l > 2 or key not in w:
            continue
        alg = 12.0 * nnz[l] + 28.0 * LEVEL_ROWS[l]
        fr = sum(f[key]) / len(f[key]) * 1024
        wr = sum(w[key]) / len(w[key]) * 1024
        tr = 2 * fr + wr
        fb = format_bytes(kind, l, words)
        lines.append(f"| {name} | {l} | {len(f[key])} | {fr / 1e6:.1f} | {2 * fr / 1e6:.1f} | {wr / 1e6:.1f} | "
                     f"{tr / 1e6:.1f} | {'%.1f' % (fb / 1e6) if fb else '-'} | {alg / 1e6:.1f} | {tr / alg:.3f} |")
        res[f"{name}@L{l}"] = {"fetch_raw_bytes": fr, "read_bytes_corrected": 2 * fr, "write_bytes": wr,
                               "traffic_bytes": tr, "format_bytes": fb, "algorithmic_bytes": alg}
    lines += ["", "Template arguments: sell_kernel<mode, 16-bit columns, non-temporal>; dict_kernel<mode, code words per row,",
              "entries decoded per row, non-temporal, rows per lane>; mode 1 = Jacobi sweep, 0 = residual, 3 = rss terms;",
              "dict_resid_restrict_kernel / dict_jacobi_prolong_kernel<code words, entries, non-temporal, rows per lane> are the",
              "fused forms (residual + restriction + first coarse sweep; last sweep + prolongation into the finer level).",
              "`format MB` = what the layout must move per launch (codes + f + x + out, plus the transfer operands of the fused",
              "forms); `algorithmic MB` = the CSR-formula figure of ONE sweep on that level (12 nnz + 28 n, SURVEY 8(d)), the",
              "figure `roofline.achieved` is quoted on for the level-0 Jacobi sweep.", ""]
    open(f"profiles/{tag}_pmc_traffic.md", "w").write("\n".join(lines))
    json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), FETCH_SIZE x2 gfx950 correction",
               "n": 4096, "layout": note, "kernels": res}, open(f"profiles/{tag}_pmc_traffic.json", "w"), indent=1)
    print("\n".join(lines))


if __name__ == "__main__":
    main()
