#!/bin/bash
# usage (GPU box, repo root): tools/levels.sh <tag> [bench args] -- rocprofv3 kernel trace of bench.py,
# per-kernel / per-level table -> gpurun_out/<tag>_by_level.md, and its patch / tail lines on stdout
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/prof_$tag -o d -- python3 $root/bench.py --steps 20 --warmup 3 --no-cpu --no-csr-ref "$@" > $root/gpurun_out/prof_$tag.log 2>&1
cd $root
python3 tools/rocprof_summary.py gpurun_out/prof_$tag/d_kernel_trace.csv > gpurun_out/${tag}_by_level.md
grep -E "patch_(down|up|rb)|band_chain" gpurun_out/${tag}_by_level.md | head -12
python3 -c "
import json
d=json.loads(open('gpurun_out/prof_$tag.log').read().strip().splitlines()[-1]); print('$tag', d['value'], 'V-cycles/s under rocprof')"
