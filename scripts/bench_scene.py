"""usage: bench_scene.py <scene> <w> <h> <spp> <depth> [rng] — renders once (after a warm-up) and prints per-kernel device times.
RTW_BENCH_NO_WARMUP=1: no warm-up render, so that every dispatch a profiler sees belongs to the one render whose per-kernel
units the JSON line states (scripts/pmc_all.sh divides summed counters by them)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracing_weekend_amd import abi
scene, w, h, spp, depth = (int(x) for x in sys.argv[1:6])
rng = int(sys.argv[6]) if len(sys.argv) > 6 else 0
r = abi.Renderer(0); r.upload_scene(abi.build_scene(scene, w, h))
if os.environ.get("RTW_BENCH_NO_WARMUP") != "1":
    r.render(abi.make_params(w, h, min(spp, 8), depth, rng_kind=rng))
img, st = r.render(abi.make_params(w, h, spp, depth, rng_kind=rng))
import zlib
print(json.dumps({"scene": scene, "lib": os.path.basename(abi.HIP_LIB), "crc": zlib.crc32(img.tobytes()), "Msamples_per_s": round(st.samples / st.seconds / 1e6, 1), "seconds": round(st.seconds, 4),
                  "samples": st.samples, "segments": st.segments, "shadow_rays": st.shadow_rays, "seg_per_sample": round(st.segments / st.samples, 3),
                  "kernels": {n: {"s": round(st.kernel_seconds[i], 4), "launches": st.kernel_launches[i], "units": st.kernel_segments[i]} for i, n in enumerate(abi.Stats.KERNELS)}}))
