#!/bin/bash
# usage (GPU box, repo root): tools/ab.sh <tag> <libA.so> <libB.so> [bench args] -- alternates A / B three
# times (same process order effects, DVFS) and prints value + dominant-kernel time of each run
tag=$1; A=$2; B=$3; shift 3
for i in 1 2 3; do
  for v in A B; do
    lib=$A; [ $v = B ] && lib=$B
    AMG_HIP_LIBRARY=$PWD/$lib python bench.py --steps 30 --warmup 5 --no-cpu --no-csr-ref "$@" > gpurun_out/ab_${tag}_$v$i.json 2> gpurun_out/ab_${tag}_$v$i.err
    python3 - <<PY
import json
try:
    d = json.load(open("gpurun_out/ab_${tag}_$v$i.json")); r = d.get("roofline") or {}
    print("$v$i", "%.1f V-cycles/s" % d["value"], r.get("kernel"), "avg %.4f min %.4f ms" % (r.get("avg_launch_ms", 0), r.get("min_launch_ms", 0)))
except Exception as e:
    print("$v$i failed", e)
PY
  done
done
