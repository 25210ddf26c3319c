#!/bin/bash
# usage: scripts/sweep_env2.sh "<A=1 B=2>" "<A=3 B=4>" ... -- <bench_scene args> : bench_scene.py once per environment string
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
ENVS=()
while [ "$1" != "--" ]; do ENVS+=("$1"); shift; done
shift
for e in "${ENVS[@]}"; do
  env $e timeout -k 10 120 python3 scripts/bench_scene.py "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$e', 'scene', d['scene'], d['crc'], d['Msamples_per_s'], {k:round(v['s'],4) for k,v in d['kernels'].items() if v['launches']})"
done
