"""Generates the committed golden fixtures from the CPU oracle (oracle/rtw_oracle.c) and the C++ host
scene description. These are THIS BUILD's reference outputs: the upstream reference holds no golden
images, known-answer vectors or tests for the path, so pixel parity against
OptiX is unpinned; what is pinned here is oracle == fixture == GPU.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from raytracing_weekend_amd import abi  # noqa: E402
import oracle  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
CASES = {
    # BASELINE.json configs[0]: Cornell box 200x200, 16 spp, depth 4 (both generators)
    "cornell_200x200_16spp_d4_philox": dict(scene=0, w=200, h=200, spp=16, depth=4, rng=0),
    "cornell_200x200_16spp_d4_lcg": dict(scene=0, w=200, h=200, spp=16, depth=4, rng=1),
    # small cases of the other in-scope scenes (BVH + moving spheres; volumes)
    "spheres_96x64_4spp_d8_philox": dict(scene=1, w=96, h=64, spp=4, depth=8, rng=0),
    "fog_96x96_8spp_d12_lcg": dict(scene=3, w=96, h=96, spp=8, depth=12, rng=1),
}

for name, c in CASES.items():
    blob = abi.build_scene(c["scene"], c["w"], c["h"])
    p = abi.make_params(c["w"], c["h"], c["spp"], c["depth"], rng_kind=c["rng"])
    img, st = oracle.render(blob, p, threads=8)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), rgb=img[..., :3].copy(), blob=np.frombuffer(blob, dtype=np.uint8).copy(),
                        meta=np.array([c["scene"], c["w"], c["h"], c["spp"], c["depth"], c["rng"], p.seed], dtype=np.int64),
                        stats=np.array([st.samples, st.segments, st.shadow_rays], dtype=np.int64))
    print(name, "segments", st.segments, "shadow", st.shadow_rays, "mean", float(img[..., :3].mean()))
