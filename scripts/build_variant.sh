#!/bin/bash
# usage: scripts/build_variant.sh <name> [extra hipcc flags...] -> raytracing_weekend_amd/csrc/variants/librtw_<name>.so
# (experiment binaries: git-ignored, they travel to the GPU box; select one with RTW_HIP_LIB=<path>)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
N=$1; shift
mkdir -p $R/raytracing_weekend_amd/csrc/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -DRTW_MIN_WAVES=4 -fPIC -shared -Wall -Wno-unused-function \
  -Rpass-analysis=kernel-resource-usage "$@" -o $R/raytracing_weekend_amd/csrc/variants/librtw_$N.so $R/raytracing_weekend_amd/csrc/rtw_hip.hip 2>&1 \
  | grep -E "error|k_pathILi0ELi0" -A8 | grep -E "error|VGPRs:|SGPRs:|Scratch|Occupancy" | sed 's/.*remark: *//; s/ \[-Rpass.*//' | tr '\n' ' '
echo " <- $N"
