#!/bin/bash
# usage (on the GPU box, through gpurun): bash scripts/profile_round2.sh r02 -> gpurun_out/profile_<tag>/
#   1. rocprofv3 --kernel-trace --stats of the default bench command (kernel_stats.csv + the JSON line of that run)
#   2. rocprofv3 --pmc passes (one per counter group, never combined with tracing) of k_path on the metric scene
#   3. a plain bench.py run (with cpu_baseline) and scripts/bench_configs.py
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
TAG=${1:-r02}
OUT=$R/gpurun_out/profile_$TAG
mkdir -p $OUT
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --no-cpu-baseline > $OUT/bench_under_rocprof.log 2>&1
cp $OUT/trace/*/*kernel_stats.csv $OUT/kernel_stats.csv
grep '^{"metric"' $OUT/bench_under_rocprof.log > $OUT/bench_under_rocprof.json || true
echo "trace done"
bash scripts/pmc_kernel.sh ${TAG}_k_path k_path 0 1920 1080 256 50 > $OUT/pmc_k_path.log 2>&1
cp gpurun_out/pmc_${TAG}_k_path/summary.json $OUT/pmc_k_path.json
echo "pmc done"
timeout -k 10 600 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
echo "bench done"
timeout -k 10 600 python3 scripts/bench_configs.py > $OUT/bench_configs.jsonl 2> $OUT/bench_configs.err
echo "configs done"
#   4. tree scenes: counters of k_trace_bvh on scene 1 (BASELINE config 3), kernel-trace statistics of scenes 1 and 4
bash scripts/pmc_kernel.sh ${TAG}_k_trace_bvh k_trace_bvh 1 1920 1080 256 50 > $OUT/pmc_k_trace_bvh.log 2>&1
cp gpurun_out/pmc_${TAG}_k_trace_bvh/summary.json $OUT/pmc_k_trace_bvh_scene1.json
bash scripts/profile_scene.sh ${TAG}_s1 1 1920 1080 512 50 > /dev/null 2>&1 && cp gpurun_out/scene_${TAG}_s1_kernel_stats.csv $OUT/kernel_stats_scene1.csv
bash scripts/profile_scene.sh ${TAG}_s4 4 1920 1080 128 50 > /dev/null 2>&1 && cp gpurun_out/scene_${TAG}_s4_kernel_stats.csv $OUT/kernel_stats_scene4.csv
echo "tree scenes done"
head -c 1500 $OUT/kernel_stats.csv; cut -c1-700 $OUT/bench.json
