"""Condenses a scripts/profile_round.sh output directory into the small files committed under profiles/."""
import csv, glob, json, os, sys, collections
out = sys.argv[1]
names = {"k_first": "k_first", "k_shade": "k_shade", "k_trace": "k_trace", "k_bounce": "k_bounce", "k_resolve": "k_resolve", "k_finish": "k_finish"}
def kind(n):
    for k in names:
        if k in n: return k
    return None
tot = collections.defaultdict(lambda: collections.defaultdict(float))
launch = collections.defaultdict(lambda: collections.defaultdict(set))
for f in glob.glob(os.path.join(out, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = kind(r["Kernel_Name"])
        if not k: continue
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        launch[k][r["Counter_Name"]].add(r["Dispatch_Id"])
summary = {}
for k in tot:
    d = {c: v for c, v in tot[k].items()}
    n = {c: len(launch[k][c]) for c in tot[k]}
    e = {"counters_sum": d, "launches": n}
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        # guide (MI355X_MICROARCH.md, HBM): FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports 1/2 of
        # the bytes of wide coalesced (16 B/lane) streaming reads -> doubled; WRITE_SIZE is exact for 16 B/lane stores
        fetch_b = d["FETCH_SIZE"] * 1024.0 * 2.0
        write_b = d["WRITE_SIZE"] * 1024.0
        e["hbm_bytes_total"] = fetch_b + write_b
        e["hbm_bytes_per_launch"] = (fetch_b + write_b) / max(1, n["FETCH_SIZE"])
        e["fetch_bytes_corrected"] = fetch_b
        e["write_bytes"] = write_b
    summary[k] = e
json.dump(summary, open(os.path.join(out, "pmc_summary.json"), "w"), indent=1)
traffic = {k + "_hbm_bytes_per_launch": round(v["hbm_bytes_per_launch"]) for k, v in summary.items() if "hbm_bytes_per_launch" in v}
traffic["note"] = ("per launch, averaged over the launches of one 128-spp render (one batch); FETCH_SIZE x2 (gfx950 wide-read correction) "
                   "+ WRITE_SIZE, KiB->B; separate --pmc passes")
json.dump(traffic, open(os.path.join(out, "pmc_traffic.json"), "w"), indent=1)
print(json.dumps(traffic, indent=1))
print(open(os.path.join(out, "kernel_stats.csv")).read()[:3000])
