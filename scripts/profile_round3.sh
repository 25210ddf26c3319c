#!/bin/bash
# usage (on the GPU box, through gpurun): bash scripts/profile_round3.sh <tag> [quick] -> gpurun_out/profile_<tag>/
#   1. rocprofv3 --kernel-trace --stats of the default bench command and of `bench.py --config c3` (kernel_stats*.csv + the JSON
#      line of the SAME run, whose roofline.avg_launch_us must agree with the CSV's AverageNs of that kernel)
#   2. rocprofv3 --pmc passes (scripts/pmc_all.sh: one pass per counter group, never combined with tracing) of every kernel of one
#      warm-up-free render of the metric scene (k_path), scene 1 (k_first / k_trace_bvh / k_shade) and scene 3 (k_path, cold)
#   3. plain bench.py runs (with cpu_baseline) of the headline and of configs c2, c3, c4
#   4. kernel-trace statistics of scenes 1, 2, 4 with the per-kernel units beside them (scripts/profile_scene.sh)
# `quick`: steps 1 (c3 only), 2 (scene 1 only) and the c3 bench line
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
TAG=${1:-r03}
QUICK=${2:-}
OUT=$R/gpurun_out/profile_$TAG
mkdir -p $OUT
trace() {  # <name> <bench args...>
  local name=$1; shift
  timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$name -- python3 bench.py --no-cpu-baseline "$@" > $OUT/bench_under_rocprof_$name.log 2>&1
  cp $OUT/trace_$name/*/*kernel_stats.csv $OUT/kernel_stats_$name.csv
  # the per-dispatch rows of the dominant kernels (start, duration): a k_path render is TWO overlapping dispatches
  python3 - $OUT/trace_$name $OUT/kernel_trace_$name.csv <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        if "k_path<" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Grid_Size_X", ""), r.get("Queue_Id", "")))
rows.sort()
if rows:
    t0 = rows[0][0]
    with open(sys.argv[2], "w") as o:
        o.write("kernel,grid_x,queue,start_ms,duration_ms\n")
        for a, b, n, g, q in rows:
            o.write('"%s",%s,%s,%.3f,%.3f\n' % (n, g, q, (a - t0) / 1e6, (b - a) / 1e6))
PY
  grep '^{"metric"' $OUT/bench_under_rocprof_$name.log > $OUT/bench_under_rocprof_$name.json || true
  rm -rf $OUT/trace_$name
  echo "trace $name done"
}
pmc() {  # <name> <bench_scene args...>
  local name=$1; shift
  bash scripts/pmc_all.sh ${TAG}_$name "$@" > $OUT/pmc_$name.log 2>&1
  cp gpurun_out/pmcall_${TAG}_$name/summary.json $OUT/pmc_$name.json
  rm -rf gpurun_out/pmcall_${TAG}_$name/p[0-9]
  echo "pmc $name done"
}
[ -z "$QUICK" ] && trace headline
trace c3 --config c3 --steps 4
pmc scene1 1 1920 1080 256 50
if [ -z "$QUICK" ]; then
  pmc headline 0 1920 1080 256 50
  pmc scene3 3 1920 1080 256 50
  timeout -k 10 600 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench done"
  for c in c2 c4; do timeout -k 10 600 python3 bench.py --config $c --steps 4 > $OUT/bench_$c.json 2> $OUT/bench_$c.err; echo "bench $c done"; done
fi
timeout -k 10 600 python3 bench.py --config c3 --steps 6 > $OUT/bench_c3.json 2> $OUT/bench_c3.err; echo "bench c3 done"
if [ -z "$QUICK" ]; then
  for sc in "s1 1 1920 1080 512 50" "s2 2 1920 1080 256 50" "s4 4 1920 1080 128 50"; do
    set -- $sc
    bash scripts/profile_scene.sh ${TAG}_$1 $2 $3 $4 $5 $6 > /dev/null 2>&1 && cp gpurun_out/scene_${TAG}_$1_kernel_stats.csv $OUT/kernel_stats_scene$2.csv && cp gpurun_out/scene_${TAG}_$1_bench.json $OUT/scene$2_bench.json
  done
  timeout -k 10 900 python3 scripts/bench_configs.py > $OUT/bench_configs.jsonl 2> $OUT/bench_configs.err; echo "configs done"
fi
cut -c1-400 $OUT/bench_c3.json
