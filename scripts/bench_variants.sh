#!/bin/bash
# usage: scripts/bench_variants.sh <spp> <variant.so...>  — prints value/ms for each library variant (same process settings)
SPP=$1; shift
for v in "$@"; do
  RTW_HIP_LIB=$v timeout -k 10 200 python bench.py --spp $SPP --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | \
   python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', 'POOL=${RTW_POOL_PATHS:-def}', 'TAIL=${RTW_TAIL_START:-def}', d['value'], 'Msamples/s', d['ms_per_step'], 'ms', 'launches', d['roofline']['launches'], 'frac', d['roofline']['frac'])"
done
