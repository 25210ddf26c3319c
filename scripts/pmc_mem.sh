#!/bin/bash
# usage: scripts/pmc_mem.sh <tag> <kernel substring> <bench_scene.py args...>
# Memory-system counters (L2 <-> fabric requests, stalls, latencies) for one kernel; separate --pmc passes, one lane.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
TAG=$1; KERN=$2; shift 2
OUT=$R/gpurun_out/pmcm_$TAG
mkdir -p $OUT
export RTW_LANES=${RTW_LANES:-1}
i=0
for grp in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_LEVEL_sum GRBM_GUI_ACTIVE" \
           "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum" \
           "TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_ATOMIC_sum TCC_EA0_ATOMIC_LEVEL_sum" \
           "TCC_TAG_STALL_sum TCC_BUSY_sum TCC_REQ_sum TCC_ATOMIC_sum" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum" \
           "TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout -k 10 75 rocprofv3 --pmc $grp --output-format csv -d $OUT/p$i -- python3 scripts/bench_scene.py "$@" > $OUT/p$i.log 2>&1 || { echo "pass $i failed (see p$i.log)"; grep -m2 -E "error code|Could not" $OUT/p$i.log; }
  echo "pass $i done"
done
python3 - $OUT "$KERN" <<'PY'
import csv, glob, os, sys, collections, json
out, kern = sys.argv[1], sys.argv[2]
tot = collections.defaultdict(float); nd = 0
for f in sorted(glob.glob(os.path.join(out, "p*", "**", "*counter_collection.csv"), recursive=True)):
    rows = [r for r in csv.DictReader(open(f)) if kern in r["Kernel_Name"]]
    nd = max(nd, len({r["Dispatch_Id"] for r in rows}))
    for r in rows: tot[r["Counter_Name"]] += float(r["Counter_Value"])
json.dump({"kernel": kern, "dispatches": nd, "counters": dict(tot)}, open(os.path.join(out, "summary.json"), "w"), indent=1)
print("dispatches", nd)
for k, v in sorted(tot.items()): print(f"{k:45s} {v:.5g}")
PY
