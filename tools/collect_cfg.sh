#!/bin/bash
# usage (GPU box, repo root): tools/collect_cfg.sh <tag> <n> <dim> <bench.py args...>
# -> gpurun_out/profiles_<tag>/: <tag>_bench_line.json, <tag>_by_level.md, <tag>_kernel_stats.csv, <tag>_pmc_traffic.md
# (kernel trace and the two PMC passes are separate runs: gpurun refuses --pmc next to other trace domains)
set -o pipefail
tag=$1; n=$2; dim=$3; shift 3
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/profiles_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/prof_$tag -o d -- python3 $root/bench.py --no-cpu --no-csr-ref "$@" > $out/${tag}_bench_line.json 2> $out/${tag}_kernel_trace.log || exit 1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $root/gpurun_out/pmc_f_$tag -o d -- python3 $root/bench.py --no-cpu --no-csr-ref "$@" --steps 3 > $out/pmc_f.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $root/gpurun_out/pmc_w_$tag -o d -- python3 $root/bench.py --no-cpu --no-csr-ref "$@" --steps 3 > $out/pmc_w.log 2>&1 || exit 1
cd $root
cp gpurun_out/prof_$tag/d_kernel_stats.csv $out/${tag}_kernel_stats.csv
python3 tools/rocprof_summary.py gpurun_out/prof_$tag/d_kernel_trace.csv > $out/${tag}_by_level.md
python3 tools/pmc_simple.py gpurun_out/pmc_f_$tag/d_counter_collection.csv gpurun_out/pmc_w_$tag/d_counter_collection.csv "$tag: bench.py $*" $n $dim > $out/${tag}_pmc_traffic.md
head -20 $out/${tag}_pmc_traffic.md
