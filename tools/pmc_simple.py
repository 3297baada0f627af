#!/usr/bin/env python3
"""HBM traffic per dispatch of every kernel of a bench.py run, from two rocprofv3 PMC passes
(--pmc FETCH_SIZE, --pmc WRITE_SIZE; each with --kernel-trace only), for any configuration
(tools/pmc_traffic.py knows the 4096^2 true-Jacobi hierarchy by heart; this one does not
classify levels, it groups by (kernel, grid size) and adds the bytes a launch has to move for
the kernels whose rows follow from the grid: dictionary sweeps and the K-Patch forms).
usage: pmc_simple.py <fetch_counter_collection.csv> <write_counter_collection.csv> <title> [n dim]
On gfx950 FETCH_SIZE counts 128-B requests at 64 B (MI355X_MICROARCH.md, HBM): reads = 2 x FETCH_SIZE."""
import csv
import re
import sys
from collections import defaultdict


def load(path):
    g = defaultdict(list)
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].split("(")[0].replace("void amg_hip::", "").replace("amg_hip::", "")
        g[(name, int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
    return g


def must_move(name, grid, levels):
    """(rows, bytes) a launch has to move, or (None, None)"""
    m = re.match(r"dict_kernel<(\d+), (\d+), \d+, \w+, (\d+)>", name)
    if m:
        rows = grid * int(m.group(3))
        return rows, rows * 25                       # 1 B row type + f + x + out
    m = re.match(r"patch_(down|up|rb)_kernel<", name)
    if m:
        for (n, pitch) in levels:
            lines = (n + pitch - 1) // pitch
            if pitch >= 128 and ((lines + 41) // 42) * (pitch // 64) * 432 == grid:
                nH = (n + 1) // 2 - 1
                extra = {"down": 16 * nH, "up": 8 * nH, "rb": 0}[m.group(1)]
                if m.group(1) == "rb" and ", true>" in name:   # the tail form also writes f_H and zeroes u_H
                    extra = 16 * nH
                elif m.group(1) == "rb" and ", true, false>" in name:   # the prolonging form also reads u_H
                    extra = 8 * nH
                return n, 25 * n + extra
    return None, None


def main():
    f, w, title = load(sys.argv[1]), load(sys.argv[2]), sys.argv[3]
    levels = []
    if len(sys.argv) > 5:
        n, dim = int(sys.argv[4]), int(sys.argv[5])
        rows, pitch = n ** dim, n
        while rows > 1000 and pitch >= 1:
            levels.append((rows, pitch))
            rows, pitch = (rows + 1) // 2 - 1, pitch // 2
    print(f"# rocprofv3 PMC traffic per dispatch: {title}\n")
    print("Two separate passes (`--pmc FETCH_SIZE --kernel-trace`, `--pmc WRITE_SIZE --kernel-trace`); reads = 2 x FETCH_SIZE")
    print("(gfx950 counts 128-B requests at 64 B), WRITE_SIZE as reported.  `must move` = what the launch reads and")
    print("writes once in the layout it streams (dictionary sweeps: 25 B per row; K-Patch legs: 25 n + 16 n_H down")
    print("-- the coarse diagonal is a kernel argument under interior tiles --, 25 n + 8 n_H up; multicolour patch stages")
    print("patch_rb_kernel<slots, mask, nt, prolong, tail>: 25 n, + 8 n_H prolonging, + 16 n_H with the residual +")
    print("restriction), where the rows follow from the grid.\n")
    print("| kernel | grid (threads) | calls | reads MB | writes MB | traffic MB | rows | must move MB | traffic / must move |")
    print("|---|---|---|---|---|---|---|---|---|")
    keys = sorted((k for k in f if k in w), key=lambda k: -(2 * sum(f[k]) + sum(w[k])))
    for k in keys[:28]:
        fr = sum(f[k]) / len(f[k]) * 1024
        wr = sum(w[k]) / len(w[k]) * 1024
        tr = 2 * fr + wr
        rows, mm = must_move(k[0], k[1], levels)
        tail = f"{rows} | {mm / 1e6:.1f} | {tr / mm:.2f}" if mm else " | | "
        print(f"| {k[0]} | {k[1]} | {len(f[k])} | {2 * fr / 1e6:.1f} | {wr / 1e6:.1f} | {tr / 1e6:.1f} | {tail} |")


if __name__ == "__main__":
    main()
