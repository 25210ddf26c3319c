#!/bin/bash
# usage: scripts/run_variants.sh <scene> <w> <h> <spp> <depth> : bench_scene.py once per library in csrc/variants (+ the in-tree one)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for lib in raytracing_weekend_amd/csrc/librtw_hip.so raytracing_weekend_amd/csrc/variants/*.so; do
  RTW_HIP_LIB=$R/$lib timeout -k 10 120 python3 scripts/bench_scene.py "$@" 2>/dev/null | cut -c1-330
done
