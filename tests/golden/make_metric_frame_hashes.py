"""usage: python tests/golden/make_metric_frame_hashes.py <out.json> [--gpu] [--threads N] [config ...]
Generates tests/golden/metric_frames.json: the CPU oracle's render of WHOLE BASELINE frames at their full sample counts
(headline: Cornell 1920x1080 at 4096 spp = 8.49 G samples; c4: Cornell + fog 1920x1080 at 2048 spp; c2: Cornell 800x800 at
1024 spp), reduced to a SHA-256 of the float32 RGBA image bytes (row-major, bottom row first, as rtw_render returns them) and
the oracle's sample / segment / probe counts. tests/test_gpu_round3.py renders the same frames through the C ABI and compares
hash and counts: every pixel of the metric workload itself is then pinned to the oracle in every `-m gpu` run, at the cost
of one render. The oracle needs ~10 minutes of a 64-thread host for the three frames (it works in bands of rows and prints
one line per band); --gpu also renders each frame on the GPU first and compares every band as it arrives, so a mismatch is
reported by pixel and not only by hash. Test infrastructure: nothing in the product imports this."""
import hashlib
import json
import os
import sys
import time
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import oracle
from raytracing_weekend_amd import abi

CONFIGS = {  # name: scene, width, height, spp, depth (BASELINE.json; bench.py CONFIGS)
    "c1": (0, 200, 200, 16, 4),
    "c2": (0, 800, 800, 1024, 50),
    "c4": (3, 1920, 1080, 2048, 50),
    "headline": (0, 1920, 1080, 4096, 50),
    # the tree scenes' whole frames at the sample counts the oracle affords (it walks no tree: 0.2 - 0.8 Msamples/s on 64 threads;
    # BASELINE config 3's own 512 spp would take it 25 minutes): several summation blocks and both streams of the wavefront pipeline
    "c3_64spp": (1, 1920, 1080, 64, 50),
    "scene2_32spp": (2, 1920, 1080, 32, 50),
    "scene4_16spp": (4, 1920, 1080, 16, 50),
}
BAND = 24


def main():
    args = sys.argv[1:]
    out_path = args.pop(0)
    use_gpu = "--gpu" in args
    if use_gpu:
        args.remove("--gpu")
    threads = 64
    if "--threads" in args:
        i = args.index("--threads")
        threads = int(args[i + 1])
        del args[i:i + 2]
    names = args or ["c1", "c2", "c4", "headline"]
    result = json.load(open(out_path)) if os.path.exists(out_path) else {}
    gpu = abi.Renderer(0) if use_gpu else None
    for name in names:
        scene, w, h, spp, depth = CONFIGS[name]
        blob = abi.build_scene(scene, w, h)
        img = st_gpu = None
        if gpu is not None:
            gpu.upload_scene(blob)
            img, st_gpu = gpu.render(abi.make_params(w, h, spp, depth))
        sha, crc = hashlib.sha256(), 0
        tot = [0, 0, 0]
        bad = 0
        t0 = time.time()
        for r0 in range(0, h, BAND):
            r1 = min(h, r0 + BAND)
            ref, st = oracle.render(blob, abi.make_params(w, h, spp, depth, row0=r0, row1=r1), threads=threads)
            b = np.ascontiguousarray(ref).tobytes()
            sha.update(b)
            crc = zlib.crc32(b, crc)
            tot[0] += st.samples; tot[1] += st.segments; tot[2] += st.shadow_rays
            if img is not None and not np.array_equal(img[r0:r1], ref):
                d = np.argwhere(np.any(img[r0:r1] != ref, axis=-1))
                bad += len(d)
                print(f"MISMATCH {name} rows {r0}-{r1}: {len(d)} pixels, first {d[:3].tolist()}", flush=True)
            print(f"{name} rows {r0}-{r1} of {h} done, {time.time() - t0:.0f} s, mismatching pixels so far {bad}", flush=True)
        entry = {"scene": scene, "width": w, "height": h, "spp": spp, "max_depth": depth, "seed": 0x6314759, "rng": "philox4x32-10",
                 "sha256": sha.hexdigest(), "crc32": crc, "samples": tot[0], "segments": tot[1], "shadow_rays": tot[2],
                 "oracle_seconds": round(time.time() - t0, 1), "oracle_threads": threads}
        if img is not None:
            entry["gpu_compared_bandwise"] = True
            entry["gpu_mismatching_pixels"] = bad
            entry["gpu_sha256"] = hashlib.sha256(np.ascontiguousarray(img).tobytes()).hexdigest()
            entry["gpu_counts_equal"] = (st_gpu.samples, st_gpu.segments, st_gpu.shadow_rays) == tuple(tot)
        result[name] = entry
        json.dump(result, open(out_path, "w"), indent=1)
        print("wrote", name, entry, flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
