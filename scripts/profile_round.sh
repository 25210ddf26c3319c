#!/bin/bash
# usage (on the GPU box): bash scripts/profile_round.sh r01   -> gpurun_out/profile_<tag>/ : kernel-trace stats + PMC traffic passes
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
TAG=${1:-r01}
OUT=$R/gpurun_out/profile_$TAG
mkdir -p $OUT
# 1. per-kernel time of the default bench command (full metric workload)
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --no-cpu-baseline > $OUT/bench_under_rocprof.log 2>&1
cp $OUT/trace/*/*kernel_stats.csv $OUT/kernel_stats.csv
# 2. HBM traffic: separate --pmc passes (FETCH_SIZE and WRITE_SIZE do not fit one pass), one 256-spp render = one batch
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --spp 128 --steps 1 --warmup 0 --no-cpu-baseline > $OUT/pmc_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py --spp 128 --steps 1 --warmup 0 --no-cpu-baseline > $OUT/pmc_write.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -- python3 bench.py --spp 128 --steps 1 --warmup 0 --no-cpu-baseline > $OUT/pmc_sq.log 2>&1
python3 scripts/profile_summarize.py $OUT
