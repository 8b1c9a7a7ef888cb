#!/bin/bash
# Counter passes over the prefill kernel (S = 2048 x 16, d = 128, causal): one rocprofv3 --pmc run per counter group, no trace
# domains beside them (gpurun refuses that combination).  Run through gpurun, then: python tools/prefill_pmc_summary.py <tag>
TAG=${1:-r04}
export ATOMA_PREFILL_CFG=${2:-4}          # 0 = prefill_mfma_kernel, 4 = the hand-scheduled prefill_asm_kernel
REPO=$(pwd)
OUT=$REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
export ATOMA_BENCH_PREFILL_SHAPE=2048x16x128
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/pfprof_$TAG -o prefill -- python $REPO/tools/bench_kernels.py prefill > $OUT/pfprof_$TAG.log 2>&1
for grp in "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_WAVES SQ_LDS_IDX_ACTIVE" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM" "SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INST_CYCLES_SALU"; do
    name=$(echo $grp | tr ' ' '_')
    timeout 300 rocprofv3 --pmc $grp --output-format csv -d $OUT/pfpmc_${TAG}_$name -o prefill -- python $REPO/tools/bench_kernels.py prefill > $OUT/pfpmc_${TAG}_$name.log 2>&1
done
ls $OUT | grep pfp
