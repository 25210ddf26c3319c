#!/bin/bash
# usage: scripts/pmc_kernel.sh <tag> <kernel substring> <scene> <w> <h> <spp> <depth>
# SQ / TCC counters of one kernel (its largest dispatch of the run), one rocprofv3 --pmc pass per counter group
# (never combined with tracing: gpurun refuses that). Writes gpurun_out/pmc_<tag>/summary.json.
set -e
export TMPDIR=/tmp
# one k_path dispatch per render (no concurrent end-game launch), so that a dispatch's counters cover all the segments counted
export RTW_PATH_FINE_BLOCKS=0
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
TAG=$1; KERN=$2; shift 2
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM" \
           "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_BRANCH" \
           "SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_IFETCH SQ_IFETCH_LEVEL SQ_LEVEL_WAVES SQ_CYCLES" \
           "SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT64 SQ_LDS_BANK_CONFLICT" \
           "GRBM_GUI_ACTIVE FETCH_SIZE" "GRBM_GUI_ACTIVE WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $grp --output-format csv -d $OUT/p$i -- python3 scripts/bench_scene.py "$@" > $OUT/p$i.log 2>&1
  echo "pass $i done: $(tail -c 300 $OUT/p$i.log | tr '\n' ' ' | cut -c1-200)"
done
python3 scripts/pmc_kernel_summary.py $OUT "$KERN"
