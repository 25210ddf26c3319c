"""Fixed cost of a render call: device seconds of the 1/8 interleaved shard and of the full metric frame over a range of sample
counts; a least-squares line seconds = a + b * spp gives the per-call constant a."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from raytracing_weekend_amd import abi
W, H, D = 1920, 1080, 50
r = abi.Renderer(0); r.upload_scene(abi.build_scene(0, W, H))
for name, kw in (("1/8 shard", dict(row0=0, row1=H, row_stride=8)), ("full frame", dict())):
    xs, ys = [], []
    for spp in (64, 128, 256, 512, 1024, 2048, 4096):
        p = abi.make_params(W, H, spp, D, **kw)
        r.render(p)
        ts = [r.render(p)[1].seconds for _ in range(3)]
        xs.append(spp); ys.append(min(ts))
    a = np.polyfit(xs, ys, 1)
    print(name, "seconds", [round(y * 1e3, 3) for y in ys], "ms; fit: %.3f ms + %.5f ms/spp" % (a[1] * 1e3, a[0] * 1e3))
