#!/bin/bash
# usage: sweep_units.sh <config> "<env>" ... : bench.py value per setting
CFG=$1; shift
for cfg in "$@"; do
  v=$(env $cfg python3 bench.py --config $CFG --steps 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{\"metric\"'):
        d=json.loads(l); print('%.0f Msamples/s  %.2f ms' % (d['value'], d['ms_per_step']))")
  echo "$CFG  $cfg  -> $v"
done
