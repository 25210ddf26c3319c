#!/bin/bash
# usage (on the GPU box): bash scripts/profile_scene.sh <tag> <scene> <w> <h> <spp> <depth>
# rocprofv3 kernel-trace statistics of one scripts/bench_scene.py run -> gpurun_out/scene_<tag>_kernel_stats.csv, and the run's
# own JSON line (per-kernel seconds, launches and UNITS: what turns the CSV's durations into roofline fractions) beside it
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
TAG=$1; shift
OUT=$R/gpurun_out/scene_$TAG
mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 scripts/bench_scene.py "$@" > $OUT/run.log 2>&1
cp $OUT/trace/*/*kernel_stats.csv $R/gpurun_out/scene_${TAG}_kernel_stats.csv
grep '^{"scene"' $OUT/run.log | tail -1 > $R/gpurun_out/scene_${TAG}_bench.json
test -s $R/gpurun_out/scene_${TAG}_bench.json || { echo "no JSON line in $OUT/run.log"; exit 1; }
head -8 $R/gpurun_out/scene_${TAG}_kernel_stats.csv | cut -c1-160
