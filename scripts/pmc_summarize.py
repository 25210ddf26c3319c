"""Summarise rocprofv3 --pmc csv passes: per kernel name, sum of each counter and launch count."""
import csv, glob, json, os, sys, collections
out = sys.argv[1]
tot = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(set))
for f in glob.glob(os.path.join(out, "p*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        k = "k_bounce_first" if "k_bounce" in k and "true" in k else "k_bounce" if "k_bounce" in k else k.split("(")[0][-40:]
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k][r["Counter_Name"]].add(r["Dispatch_Id"])
res = {}
for k in tot:
    res[k] = {c: {"sum": v, "launches": len(cnt[k][c])} for c, v in tot[k].items()}
json.dump(res, open(os.path.join(out, "summary.json"), "w"), indent=1)
for k in res:
    if "k_bounce" not in k: continue
    print(k)
    for c, v in sorted(res[k].items()):
        print(f"   {c:28s} {v['sum']:.6g}  (launches {v['launches']})")
