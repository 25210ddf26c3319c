"""Summarise rocprofv3 --pmc csv passes per k_bounce dispatch (in launch order) and in total."""
import csv, glob, json, os, sys, collections
out = sys.argv[1]
per = collections.defaultdict(dict)   # dispatch order index -> counter -> value
names = {}
for f in sorted(glob.glob(os.path.join(out, "p*", "**", "*counter_collection.csv"), recursive=True)):
    rows = [r for r in csv.DictReader(open(f)) if "k_bounce" in r["Kernel_Name"]]
    ids = sorted({int(r["Dispatch_Id"]) for r in rows})
    order = {d: i for i, d in enumerate(ids)}
    for r in rows:
        i = order[int(r["Dispatch_Id"])]
        per[i][r["Counter_Name"]] = per[i].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        names[i] = "first" if "true" in r["Kernel_Name"] else "bounce"
res = {str(i): {"kernel": names[i], **per[i]} for i in sorted(per)}
json.dump(res, open(os.path.join(out, "summary.json"), "w"), indent=1)
cols = ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_ANY",
        "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "GRBM_GUI_ACTIVE", "FETCH_SIZE", "WRITE_SIZE"]
print("i kind " + " ".join(c.replace("SQ_", "") for c in cols) + " lane_util active% wait%")
for i in sorted(per):
    p = per[i]
    lu = p.get("SQ_THREAD_CYCLES_VALU", 0) / max(1.0, p.get("SQ_ACTIVE_INST_VALU", 0) * 64)
    wc = max(1.0, p.get("SQ_WAVE_CYCLES", 0))
    print(i, names[i], " ".join(f"{p.get(c, 0):.4g}" for c in cols), f"{lu:.3f} {p.get('SQ_ACTIVE_INST_ANY', 0) / wc:.3f} {p.get('SQ_WAIT_ANY', 0) / wc:.3f}")
