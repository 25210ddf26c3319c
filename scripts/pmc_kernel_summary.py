"""Summarise the rocprofv3 --pmc passes of scripts/pmc_kernel.sh: counters of the LARGEST dispatch (by SQ_WAVE_CYCLES or
first counter seen) of the kernels whose name contains the given substring, plus derived ratios."""
import collections, csv, glob, json, os, sys
out, kern = sys.argv[1], sys.argv[2]
tot = {}
for d in sorted(glob.glob(os.path.join(out, "p*"))):
    if not os.path.isdir(d):
        continue
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if kern in r["Kernel_Name"]:
                per[int(r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
    if not per:
        continue
    # the largest dispatch of this pass (all counters of a pass scale together)
    big = max(per, key=lambda k: sum(per[k].values()))
    for c, v in per[big].items():
        tot[c] = v
# the profiled command's own JSON line (scripts/bench_scene.py): units = radiance segments the kernel processed in the main render
units = None
for lg in sorted(glob.glob(os.path.join(out, "p*.log"))):
    for line in open(lg, errors="ignore"):
        if line.startswith("{") and '"kernels"' in line:
            try:
                kk = json.loads(line)["kernels"]
                for name, v in kk.items():
                    if name in kern or kern in name:
                        units = v["units"]
            except Exception:
                pass
def r(a, b):
    return tot.get(a, 0.0) / tot[b] if tot.get(b) else None
der = {
    "valu_per_wave": r("SQ_INSTS_VALU", "SQ_WAVES"), "salu_per_wave": r("SQ_INSTS_SALU", "SQ_WAVES"), "smem_per_wave": r("SQ_INSTS_SMEM", "SQ_WAVES"),
    "lane_utilisation": (tot["SQ_THREAD_CYCLES_VALU"] / (64.0 * tot["SQ_ACTIVE_INST_VALU"])) if tot.get("SQ_ACTIVE_INST_VALU") else None,
    "frac_wave_cycles_issuing_any": r("SQ_ACTIVE_INST_ANY", "SQ_WAVE_CYCLES"), "frac_wave_cycles_waiting": r("SQ_WAIT_ANY", "SQ_WAVE_CYCLES"),
    "frac_wave_cycles_issue_stalled": r("SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES"), "frac_wave_cycles_valu": r("SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES"),
    "smem_latency_cycles": r("SQ_INST_LEVEL_SMEM", "SQ_INSTS_SMEM"), "lds_latency_cycles": r("SQ_INST_LEVEL_LDS", "SQ_INSTS_LDS"),
    "ifetch_latency": r("SQ_IFETCH_LEVEL", "SQ_IFETCH"), "waves_per_sq_cycle": r("SQ_LEVEL_WAVES", "SQ_CYCLES"),
    "hbm_bytes": (2.0 * tot.get("FETCH_SIZE", 0.0) + tot.get("WRITE_SIZE", 0.0)) * 1024.0 if ("FETCH_SIZE" in tot or "WRITE_SIZE" in tot) else None,
}
if units:
    der["segments"] = units
    der["hbm_bytes_per_segment"] = der["hbm_bytes"] / units if der.get("hbm_bytes") is not None else None
    der["valu_insts_per_segment"] = tot.get("SQ_INSTS_VALU", 0.0) / units
    der["salu_insts_per_segment"] = tot.get("SQ_INSTS_SALU", 0.0) / units
json.dump({"kernel": kern, "counters": tot, "derived": der}, open(os.path.join(out, "summary.json"), "w"), indent=1)
print(json.dumps(der, indent=1))
print(" ".join(f"{k.replace('SQ_', '')}={v:.4g}" for k, v in sorted(tot.items())))
