#!/bin/bash
# usage: scripts/sweep_scene.sh "<bench_scene args>" "<env settings>" ... : median Msamples/s of 5 renders per setting (one process each)
ARGS=$1; shift
for cfg in "$@"; do
  for rep in 1 2 3 4 5; do env $cfg python3 scripts/bench_scene.py $ARGS 2>/dev/null; done | grep '^{"scene' | python3 -c "
import sys,json
v=sorted(json.loads(l)['Msamples_per_s'] for l in sys.stdin)
print('%-50s median %.0f  (min %.0f max %.0f)' % ('$cfg', v[len(v)//2], v[0], v[-1]))"
done
