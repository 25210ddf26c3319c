#!/bin/bash
# usage: scripts/ab_variants.sh <variant> [<variant> ...] -- "<bench_scene args>" ["<bench_scene args>" ...]
# bench_scene.py once per (variant library, scene); a variant is a name under csrc/variants (librtw_<name>.so), "tree" for
# the in-tree library, and may carry environment settings in front: "RTW_PATH_TREE=1 base". One compact line per run; the
# CRC of the image shows at once when a variant changes a bit.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
VARS=()
while [ "$1" != "--" ]; do VARS+=("$1"); shift; done
shift
for sc in "$@"; do
  for v in "${VARS[@]}"; do
    name=${v##* }
    envs=""
    if [ "$name" != "$v" ]; then envs=${v% *}; fi
    lib=raytracing_weekend_amd/csrc/variants/librtw_$name.so
    if [ "$name" = "tree" ]; then lib=raytracing_weekend_amd/csrc/librtw_hip.so; fi
    env $envs RTW_HIP_LIB=$R/$lib timeout -k 10 120 python3 scripts/bench_scene.py $sc 2>/dev/null \
      | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-28s' % '$v', 'scene', d['scene'], 'crc', d['crc'], d['Msamples_per_s'], {k:round(v['s'],4) for k,v in d['kernels'].items() if v['launches']})"
  done
done
