#!/usr/bin/env python3
"""Condenses a rocprofv3 --kernel-trace CSV into a per-(kernel, grid) table.

The V-cycle launches the same kernel template on every level, so the per-name
averages of `--stats` mix 16M-row and 4K-row launches.  Grouping by grid size
separates the levels; the fine-level rows are what bench.py's roofline quotes.

usage: rocprof_summary.py <kernel_trace.csv> [bytes_per_fine_sweep]
"""
import csv
import sys
from collections import defaultdict


def short(name):
    name = name.replace("amg_hip::", "").replace("void ", "")
    i = name.find("(")
    return name if i < 0 else name[:i]


def main():
    path = sys.argv[1]
    sweep_bytes = float(sys.argv[2]) if len(sys.argv) > 2 else None
    groups = defaultdict(list)
    meta = {}
    with open(path) as fh:
        for row in csv.DictReader(fh):
            key = (short(row["Kernel_Name"]), int(row["Grid_Size_X"]), int(row["Workgroup_Size_X"]))
            groups[key].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
            meta[key] = (row["VGPR_Count"], row["SGPR_Count"], row["LDS_Block_Size"])
    total = sum(sum(v) for v in groups.values())
    print("| kernel | grid (threads) | block | calls | avg us | min us | max us | % time | VGPR | LDS B |")
    print("|---|---|---|---|---|---|---|---|---|---|")
    for key, v in sorted(groups.items(), key=lambda kv: -sum(kv[1])):
        name, grid, wg = key
        avg = sum(v) / len(v) / 1e3
        line = (f"| {name} | {grid} | {wg} | {len(v)} | {avg:.1f} | {min(v) / 1e3:.1f} | "
                f"{max(v) / 1e3:.1f} | {100.0 * sum(v) / total:.1f} | {meta[key][0]} | {meta[key][2]} |")
        print(line)
    if sweep_bytes:
        print()
        for key, v in sorted(groups.items(), key=lambda kv: -sum(kv[1])):
            if ("csr_stage_kernel<1" in key[0] or "sell_kernel<1" in key[0]) and len(v) > 4:
                big = max(k[1] for k in groups if k[0] == key[0])
                if key[1] == big:
                    avg = sum(v) / len(v)
                    print(f"fine-level Jacobi sweep: {key[0]} grid {key[1]}: avg {avg / 1e3:.1f} us over "
                          f"{len(v)} launches -> {sweep_bytes / avg:.0f} GB/s algorithmic "
                          f"({sweep_bytes / avg / 80.0:.1f}% of 8 TB/s)")
                    break


if __name__ == "__main__":
    main()
