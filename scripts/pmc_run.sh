#!/bin/bash
# usage: scripts/pmc_run.sh <tag> <bench args...>   (run on the GPU box; one rocprofv3 --pmc pass per counter group)
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
TAG=$1; shift
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_IFETCH SQ_INST_CYCLES_SMEM SQ_INSTS_VMEM" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $grp --output-format csv -d $OUT/p$i -- python3 bench.py "$@" --no-cpu-baseline > $OUT/p$i.log 2>&1
  echo "pass $i done"
done
python3 scripts/pmc_summarize.py $OUT
