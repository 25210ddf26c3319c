#!/bin/bash
# usage: scripts/sweep_shard.sh "<env settings>" ... : median device seconds of the 1/8 interleaved shard of the metric frame per setting
for cfg in "$@"; do
  for rep in 1 2; do env $cfg python3 scripts/bench_shard.py 8 4096 6; done | grep '^{"lib' | python3 -c "
import sys,json
v=[]
for l in sys.stdin: v+=json.loads(l)['seconds']
v=sorted(v); print('%-60s min %.5f median %.5f' % ('$cfg', v[0], v[len(v)//2]))"
done
