#!/bin/bash
# usage (on the GPU box, from the repo root): tools/prof.sh <tag> [bench.py args...]
# rocprofv3 kernel trace of bench.py -> gpurun_out/prof_<tag>/ + per-level summary
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/prof_$tag -o d -- python3 $root/bench.py "$@" > $root/gpurun_out/prof_$tag.log 2>&1
cd $root
python3 tools/rocprof_summary.py gpurun_out/prof_$tag/d_kernel_trace.csv > gpurun_out/${tag}_by_level.md
