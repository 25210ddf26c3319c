#!/bin/bash
# usage: scripts/pmc_scene.sh <tag> <kernel substring> <bench_scene.py args...>
# Per-kernel SQ/TCP counter totals for one scene render (separate --pmc passes; one stream lane so kernels do not overlap).
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
TAG=$1; KERN=$2; shift 2
OUT=$R/gpurun_out/pmcs_$TAG
mkdir -p $OUT
export RTW_LANES=1
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
           "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $grp --output-format csv -d $OUT/p$i -- python3 scripts/bench_scene.py "$@" > $OUT/p$i.log 2>&1 || echo "pass $i failed (see p$i.log)"
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
def g(k): return tot.get(k, 0.0)
def r(a, b): return g(a) / max(1.0, g(b))
d = {"kernel": kern, "dispatches": nd, "counters": dict(tot),
     "valu_per_wave": r("SQ_INSTS_VALU", "SQ_WAVES"), "salu_per_wave": r("SQ_INSTS_SALU", "SQ_WAVES"), "vmem_per_wave": r("SQ_INSTS_VMEM", "SQ_WAVES"),
     "lds_per_wave": r("SQ_INSTS_LDS", "SQ_WAVES"), "smem_per_wave": r("SQ_INSTS_SMEM", "SQ_WAVES"),
     "lane_util": g("SQ_THREAD_CYCLES_VALU") / max(1.0, 64 * g("SQ_ACTIVE_INST_VALU")),
     "active_any_frac": r("SQ_ACTIVE_INST_ANY", "SQ_WAVE_CYCLES"), "wait_inst_any_frac": r("SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES"),
     "vmem_latency_cyc": r("SQ_INST_LEVEL_VMEM", "SQ_INSTS_VMEM"), "lds_latency_cyc": r("SQ_INST_LEVEL_LDS", "SQ_INSTS_LDS"),
     "waves_per_sq_cycle": r("SQ_LEVEL_WAVES", "SQ_CYCLES"),
     "valu_busy_frac_of_sq": r("SQ_ACTIVE_INST_VALU", "SQ_BUSY_CYCLES")}
json.dump(d, open(os.path.join(out, "summary.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in d.items() if k != "counters"}, indent=1))
print({k: f"{v:.4g}" for k, v in tot.items()})
PY
