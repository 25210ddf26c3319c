#!/bin/bash
# usage: scripts/pmc_run2.sh <tag> <bench args...> : latency-oriented counter groups
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
TAG=$1; shift
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VMEM SQ_INST_LEVEL_VMEM SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM SQ_IFETCH SQ_IFETCH_LEVEL" \
           "SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR" \
           "SQ_BUSY_CU_CYCLES SQ_LEVEL_WAVES SQ_CYCLES SQ_INSTS_BRANCH SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $grp --output-format csv -d $OUT/p$i -- python3 bench.py "$@" --no-cpu-baseline > $OUT/p$i.log 2>&1
  echo "pass $i done"
done
python3 - $OUT <<'PY'
import csv, glob, os, sys, collections
out=sys.argv[1]
per=collections.defaultdict(dict)
for f in sorted(glob.glob(os.path.join(out,"p*","**","*counter_collection.csv"),recursive=True)):
    rows=[r for r in csv.DictReader(open(f)) if "k_bounce" in r["Kernel_Name"]]
    ids=sorted({int(r["Dispatch_Id"]) for r in rows}); order={d:i for i,d in enumerate(ids)}
    for r in rows:
        i=order[int(r["Dispatch_Id"])]; per[i][r["Counter_Name"]]=per[i].get(r["Counter_Name"],0.0)+float(r["Counter_Value"])
for i in sorted(per)[:5]:
    print(i, " ".join(f"{k.replace('SQ_','')}={v:.4g}" for k,v in sorted(per[i].items())))
    p=per[i]
    def r(a,b): return p.get(a,0)/max(1.0,p.get(b,0))
    print("   avg VMEM latency(cyc?)", r("SQ_INST_LEVEL_VMEM","SQ_INSTS_VMEM"), "SMEM", r("SQ_INST_LEVEL_SMEM","SQ_INSTS_SMEM"), "IFETCH", r("SQ_IFETCH_LEVEL","SQ_IFETCH"), "occupancy waves/SQcycle", r("SQ_LEVEL_WAVES","SQ_CYCLES"))
PY
