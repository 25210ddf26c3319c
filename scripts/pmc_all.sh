#!/bin/bash
# usage: scripts/pmc_all.sh <tag> <scene> <w> <h> <spp> <depth>
# SQ / TCC counters of EVERY path-tracing kernel of one render (no warm-up render: RTW_BENCH_NO_WARMUP=1), one rocprofv3 --pmc
# pass per counter group (never combined with tracing: gpurun refuses that), summed over all dispatches of a kernel and divided
# by the units (segments / path slots) the render's own JSON line states for that kernel -> per-dispatch-correct per-unit
# figures. Writes gpurun_out/pmcall_<tag>/summary.json (scripts/pmc_all_summary.py).
set -e
export TMPDIR=/tmp
export RTW_BENCH_NO_WARMUP=1
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
TAG=$1; shift
OUT=$R/gpurun_out/pmcall_$TAG
mkdir -p $OUT
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM" \
           "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_BRANCH" \
           "SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LEVEL_WAVES SQ_CYCLES" \
           "GRBM_GUI_ACTIVE FETCH_SIZE" "GRBM_GUI_ACTIVE WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $grp --output-format csv -d $OUT/p$i -- python3 scripts/bench_scene.py "$@" > $OUT/p$i.log 2>&1
  echo "pass $i done: $(tail -c 300 $OUT/p$i.log | tr '\n' ' ' | cut -c1-160)"
done
python3 scripts/pmc_all_summary.py $OUT
