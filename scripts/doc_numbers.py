"""usage: doc_numbers.py [profiles dir] [rNN] — prints the figures the documents quote from a round's published profiles
(bench.py line, k_path dispatches, per-scene kernel shares), for checking them by eye."""
import csv, json, os, sys
d = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
r = sys.argv[2] if len(sys.argv) > 2 else "r02"
for l in open(os.path.join(d, f"{r}_bench_configs.jsonl")):
    c = json.loads(l)
    print("%-46s %8.4f s %9.0f Msamples/s %7.0f GB/s %5.1f %% bit-exact %s" % (c["config"][:46], c["device_seconds"], c["Msamples_per_s"], c["algorithmic_GBps"], 100 * c["frac_of_8TBps"], c["band_bit_exact"]))
for n in ("bench", "bench_under_rocprof"):
    b = json.load(open(os.path.join(d, f"{r}_{n}.json")))
    print(n, "value", b["value"], "ms/step", b["ms_per_step"], "avg_launch_us", b["roofline"]["avg_launch_us"], "frac", round(b["roofline"]["frac"], 4),
          "valu", b["roofline"]["valu"]["frac"], b["roofline"]["valu"]["achieved_Ginst_per_s"], "cpu", b.get("cpu_baseline", {}).get("value"), b.get("cpu_baseline", {}).get("single_thread_value"))
p = os.path.join(d, f"{r}_kernel_trace_k_path.csv")
if os.path.exists(p):
    print("k_path dispatches (start, duration ms):", [(row["start_ms_after_first"], row["duration_ms"]) for row in csv.DictReader(open(p))])
for row in list(csv.DictReader(open(os.path.join(d, f"{r}_kernel_stats.csv"))))[:1]:
    print("kernel_stats:", row["Name"][:30], "calls", row["Calls"], "avg ms", round(float(row["AverageNs"]) / 1e6, 1))
for sc in (1, 4):
    p = os.path.join(d, f"{r}_kernel_stats_scene{sc}.csv")
    if os.path.exists(p):
        print("scene", sc, [(row["Name"][11:24], row["Percentage"]) for row in list(csv.DictReader(open(p)))[:4]])
p = os.path.join(d, f"{r}_pmc_k_trace_bvh_scene1.json")
if os.path.exists(p):
    t = json.load(open(p)); c = t["counters"]
    print("k_trace_bvh scene 1: lane utilisation %.3f, SALU/VALU %.3f, waves %d" % (t["derived"]["lane_utilisation"], c["SQ_INSTS_SALU"] / c["SQ_INSTS_VALU"], c["SQ_WAVES"]))
