#!/bin/bash
# usage (on the GPU box, from the repo root): tools/collect_profiles.sh <tag> [extra bench.py args]
# -> gpurun_out/profiles_<tag>/: <tag>_bench_kernel_stats.csv, <tag>_bench_by_level.md,
#    <tag>_pmc_traffic.{md,json}; copy them into profiles/ afterwards.
# PMC passes are separate runs with --kernel-trace only (gpurun refuses --pmc with the
# sys/hip/hsa trace domains).
set -o pipefail
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/profiles_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/prof_$tag -o d -- python3 $root/bench.py --steps 20 --warmup 3 --no-cpu --no-csr-ref "$@" > $out/${tag}_bench_line.json 2> $out/${tag}_kernel_trace.log || exit 1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $root/gpurun_out/pmc_f_$tag -o d -- python3 $root/bench.py --steps 4 --warmup 1 --no-cpu --no-csr-ref "$@" > $out/pmc_f.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $root/gpurun_out/pmc_w_$tag -o d -- python3 $root/bench.py --steps 4 --warmup 1 --no-cpu --no-csr-ref "$@" > $out/pmc_w.log 2>&1 || exit 1
cd $root
cp gpurun_out/prof_$tag/d_kernel_stats.csv $out/${tag}_bench_kernel_stats.csv
python3 tools/rocprof_summary.py gpurun_out/prof_$tag/d_kernel_trace.csv > $out/${tag}_bench_by_level.md
note="default layout (dictionary-coded rows, K-Patch on levels 0-3)"
case " $* " in *" --layout sell "*) note="--layout sell (CSR sliced into 64-row panels, 16-bit relative columns)";; esac
PMC_OUT_DIR=$out python3 tools/pmc_traffic.py gpurun_out/pmc_f_$tag/d_counter_collection.csv gpurun_out/pmc_w_$tag/d_counter_collection.csv $tag "$note" > /dev/null
grep -v "^$" $out/${tag}_pmc_traffic.md | head -30
