"""Quick GPU-vs-oracle parity probe (development aid; the real parity tests are tests/test_gpu_parity.py)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from raytracing_weekend_amd import abi
import oracle

def compare(scene, w, h, spp, depth, rng, **kw):
    blob = abi.build_scene(scene, w, h)
    p = abi.make_params(w, h, spp, depth, rng_kind=rng, **kw)
    r = abi.Renderer(0)
    r.upload_scene(blob)
    t = time.time(); img, st = r.render(p); dt = time.time() - t
    ref, sr = oracle.render(blob, p, threads=16)
    d = img[..., :3].astype(np.float64) - ref[..., :3]
    rmse = np.sqrt(np.mean(d * d))
    nbad = int((img[..., :3] != ref[..., :3]).any(axis=-1).sum())
    print(f"scene {scene} {w}x{h} spp {spp} depth {depth} rng {rng}: rmse {rmse:.3e} maxabs {np.abs(d).max():.3e} "
          f"pixels_differing {nbad}/{w*h} seg gpu {st.segments} cpu {sr.segments} shadow gpu {st.shadow_rays} cpu {sr.shadow_rays} "
          f"gpu_s {st.seconds:.4f} wall {dt:.3f}", flush=True)
    if nbad:
        ys, xs = np.nonzero((img[..., :3] != ref[..., :3]).any(axis=-1))
        for y, x in list(zip(ys, xs))[:5]:
            print("   diff at", x, y, img[y, x, :3], ref[y, x, :3])
    r.close()
    return rmse

if __name__ == "__main__":
    for rng in (0, 1):
        compare(0, 64, 64, 4, 4, rng)
        compare(0, 200, 200, 16, 4, rng)
        compare(3, 96, 96, 8, 12, rng)
        compare(1, 96, 64, 4, 8, rng)
    compare(0, 128, 128, 32, 50, 0, samples_per_pass=5)
