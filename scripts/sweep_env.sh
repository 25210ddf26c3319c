#!/bin/bash
# usage: scripts/sweep_env.sh <VAR> "<v1 v2 ...>" <bench_scene args> : bench_scene.py once per value of one environment variable
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
VAR=$1; VALS=$2; shift 2
for v in $VALS; do
  env $VAR=$v timeout -k 10 120 python3 scripts/bench_scene.py "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$VAR=$v', 'scene', d['scene'], d['crc'], d['Msamples_per_s'], {k:round(v['s'],4) for k,v in d['kernels'].items() if v['launches']})"
done
