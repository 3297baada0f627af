#!/usr/bin/env python3
"""Per-kernel averages of the SQ counters of tools/pmc_sq.sh (rocprofv3 counter_collection CSVs)."""
import collections
import csv
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for path in sys.argv[1:]:
    try:
        rows = list(csv.DictReader(open(path)))
    except OSError:
        continue
    for r in rows:
        k = r.get("Kernel_Name", "").replace("amg_hip::", "")
        k = k.split("(")[0]
        grid = r.get("Grid_Size", "")
        a = acc[(k, grid)][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"])
        a[1] += 1
names = sorted({c for v in acc.values() for c in v})
big = sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", [0, 1])[0])[:12]
print("| kernel | grid | " + " | ".join(names) + " |")
print("|---|---|" + "---|" * len(names))
for (k, grid), v in big:
    print(f"| {k[:60]} | {grid} | " + " | ".join(f"{v[c][0] / max(v[c][1], 1):.4g}" if c in v else "" for c in names) + " |")
