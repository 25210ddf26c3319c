"""Times the trace-only debug kernel (closest hit, brute lists) on Cornell-box rays: a proxy for a stand-alone trace kernel."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from raytracing_weekend_amd import abi
n = 8_000_000
rng = np.random.default_rng(1)
o = rng.uniform(10, 545, (n, 3)).astype(np.float32)
d = rng.normal(size=(n, 3)).astype(np.float32)
rays = np.concatenate([o, d, np.full((n, 1), 1e-6, np.float32), np.full((n, 1), 1e27, np.float32)], axis=1)
r = abi.Renderer(0)
r.upload_scene(abi.build_scene(0, 64, 64))
for _ in range(3):
    t0 = time.time(); t, p = r.debug_intersect(rays); print("call", time.time() - t0, (p >= 0).mean())
