"""usage: python scripts/profile_publish.py <gpurun_out/profile_TAG> <rNN> : copies the summaries of a scripts/profile_round3.sh run into
profiles/ (tracked) under the round's prefix and regenerates profiles/pmc_traffic.json - the per-UNIT traffic and instruction
figures, per configuration and kernel, that bench.py scales into roofline.traffic / roofline.valu / roofline.per_kernel."""
import json, os, shutil, sys
src, rnd = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(root, "profiles")
copied = []
for name in sorted(os.listdir(src)):
    p = os.path.join(src, name)
    if not os.path.isfile(p) or not os.path.getsize(p):
        continue
    if name.endswith((".csv", ".json", ".jsonl")):
        shutil.copy(p, os.path.join(dst, f"{rnd}_{name}"))
        copied.append(name)
print("copied", copied)


def entries(pmc_file, kernels, what):
    d = json.load(open(os.path.join(src, pmc_file)))
    out = {}
    for k in kernels:
        e = d["kernels"].get(k)
        if not e:
            continue
        r = e["derived"]
        out[k] = {"hbm_bytes_per_unit": r.get("hbm_bytes_per_unit"), "valu_insts_per_unit": r.get("valu_insts_per_unit"),
                  "salu_insts_per_unit": r.get("salu_insts_per_unit"), "lane_utilisation": r.get("lane_utilisation"),
                  "lds_bank_conflict_per_lds_active": r.get("lds_bank_conflict_per_lds_active"),
                  "units_profiled": r.get("units"), "dispatches_profiled": r.get("dispatches"),
                  "source": f"profiles/{rnd}_{pmc_file}: scripts/pmc_all.sh, separate rocprofv3 --pmc passes of `{what}` (one render, no warm-up), counters "
                            "summed over all dispatches of the kernel / the units the render states for it; HBM bytes = FETCH_SIZE x 2 (gfx950 "
                            "wide-read correction, MI355X_MICROARCH.md HBM) + WRITE_SIZE, KiB -> B"}
    return out


traffic = {}
if os.path.exists(os.path.join(src, "pmc_headline.json")):
    traffic["headline"] = entries("pmc_headline.json", ["k_path"], "scripts/bench_scene.py 0 1920 1080 256 50")
if os.path.exists(os.path.join(src, "pmc_scene1.json")):
    traffic["c3"] = entries("pmc_scene1.json", ["k_first", "k_trace", "k_shade", "k_bounce"], "scripts/bench_scene.py 1 1920 1080 256 50")
if os.path.exists(os.path.join(src, "pmc_scene3.json")):
    traffic["c4"] = entries("pmc_scene3.json", ["k_path"], "scripts/bench_scene.py 3 1920 1080 256 50")
json.dump(traffic, open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1)
json.dump(traffic, open(os.path.join(dst, f"{rnd}_pmc_traffic.json"), "w"), indent=1)
for cfg, ks in traffic.items():
    for k, v in ks.items():
        print(cfg, k, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items() if a != "source"})
