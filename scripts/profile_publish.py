"""usage: python scripts/profile_publish.py <gpurun_out/profile_TAG> <rNN> : copies the summaries of a scripts/profile_round2.sh run
into profiles/ (tracked) under the round's prefix and regenerates profiles/pmc_traffic.json, the per-segment figures bench.py
scales into roofline.traffic / roofline.valu."""
import json, os, shutil, sys
src, rnd = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(root, "profiles")
for name, out in (("kernel_stats.csv", "kernel_stats.csv"), ("bench_under_rocprof.json", "bench_under_rocprof.json"), ("pmc_k_path.json", "pmc_k_path.json"),
                  ("bench.json", "bench.json"), ("bench_configs.jsonl", "bench_configs.jsonl"),
                  ("pmc_k_trace_bvh_scene1.json", "pmc_k_trace_bvh_scene1.json"), ("kernel_stats_scene1.csv", "kernel_stats_scene1.csv"),
                  ("kernel_stats_scene4.csv", "kernel_stats_scene4.csv")):
    p = os.path.join(src, name)
    if os.path.exists(p) and os.path.getsize(p):
        shutil.copy(p, os.path.join(dst, f"{rnd}_{out}"))
pm = json.load(open(os.path.join(src, "pmc_k_path.json")))
d = pm["derived"]
traffic = {
    "k_path": {"hbm_bytes_per_segment": d["hbm_bytes_per_segment"], "valu_insts_per_segment": d["valu_insts_per_segment"],
               "salu_insts_per_segment": d["salu_insts_per_segment"], "lane_utilisation": d["lane_utilisation"], "segments_profiled": d["segments"],
               "source": f"profiles/{rnd}_pmc_k_path.json: scripts/pmc_kernel.sh, rocprofv3 --pmc passes of `scripts/bench_scene.py 0 1920 1080 256 50` "
                         "(the metric scene, one k_path launch); HBM bytes = FETCH_SIZE x 2 (gfx950 wide-read correction, MI355X_MICROARCH.md HBM) + WRITE_SIZE, KiB -> B"},
    # wavefront kernels (tree scenes, RTW_PATH=0): round-1 measurement restated per segment (profiles/r01_pmc_summary.json: bytes of all
    # launches of a 128-spp render / the segments those launches shaded)
    "k_shade": {"hbm_bytes_per_segment": 57.93e9 / 306.8e6, "source": "profiles/r01_pmc_summary.json (15 launches, 306.8 M segments)"},
    "k_first": {"hbm_bytes_per_segment": 15.11e9 / 265.4e6, "source": "profiles/r01_pmc_summary.json (3 launches, 265.4 M camera paths)"},
}
json.dump(traffic, open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1)
json.dump(traffic, open(os.path.join(dst, f"{rnd}_pmc_traffic.json"), "w"), indent=1)
print(json.dumps(traffic["k_path"], indent=1))
