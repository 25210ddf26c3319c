#!/bin/bash
# usage: scripts/run_variants_env.sh "<ENV=.. ENV=..>" <bench_scene args> : bench_scene.py per variant library with the given environment
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
E=$1; shift
for lib in raytracing_weekend_amd/csrc/variants/*.so; do
  env $E RTW_HIP_LIB=$R/$lib timeout -k 10 120 python3 scripts/bench_scene.py "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$E', d['lib'], d['crc'], d['Msamples_per_s'], {k:round(v['s'],4) for k,v in d['kernels'].items() if v['launches']})"
done
