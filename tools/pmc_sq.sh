#!/bin/bash
# usage (GPU box, repo root): tools/pmc_sq.sh <tag> [bench.py args]
# SQ counter passes (wave time split: busy / waiting / issuing) for the kernels of bench.py;
# --pmc with --kernel-trace only (gpurun refuses --pmc next to the sys / hip / hsa trace domains).
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $root/gpurun_out/sq_${tag}_$i -o d -- python3 $root/bench.py --steps 3 --warmup 1 --no-cpu --no-csr-ref "$@" > $root/gpurun_out/sq_${tag}_$i.log 2>&1 || echo "pass $i failed"
done
cd $root
python3 tools/pmc_sq_summary.py gpurun_out/sq_${tag}_*/d_counter_collection.csv > gpurun_out/sq_${tag}_summary.md
