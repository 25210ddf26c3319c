"""Times the 8 row tiles of the metric frame separately on one GPU: predicts the strong-scaling balance of the 8-GPU split."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracing_weekend_amd import abi
W, H, SPP, D = 1920, 1080, 4096, 50
r = abi.Renderer(0); r.upload_scene(abi.build_scene(0, W, H))
for n in (8,):
    rows = [(g * H) // n for g in range(n + 1)]
    ts = []
    for g in range(n):
        p = abi.make_params(W, H, SPP, D, row0=rows[g], row1=rows[g + 1])
        r.render(abi.make_params(W, H, 8, D, row0=rows[g], row1=rows[g + 1]))
        _, st = r.render(p)
        ts.append(st.seconds)
    print(n, "tiles: seconds", [round(t, 4) for t in ts], "max/mean %.3f" % (max(ts) / (sum(ts) / n)), "ideal speedup vs sum %.2f" % (sum(ts) / max(ts)))

for n in (2, 4, 8):
    ts = []
    for g in range(n):
        p = abi.make_params(W, H, SPP, D, row0=g, row1=H, row_stride=n)
        r.render(abi.make_params(W, H, 8, D, row0=g, row1=H, row_stride=n))
        _, st = r.render(p)
        ts.append(st.seconds)
    print(n, "interleaved shards: seconds", [round(t, 4) for t in ts], "max/mean %.3f" % (max(ts) / (sum(ts) / n)), "ideal speedup vs sum %.2f" % (sum(ts) / max(ts)))
_, st1 = r.render(abi.make_params(W, H, SPP, D))
print("single tile seconds", round(st1.seconds, 4))
# strong-scaling estimate from one GPU: the time of the slowest 1/n shard against 1/n of the full frame's time
import time
for n in (2, 4, 8):
    p = abi.make_params(W, H, SPP, D, row0=0, row1=H, row_stride=n)
    r.render(p)
    t0 = time.perf_counter(); _, st = r.render(p); wall = time.perf_counter() - t0
    print(n, "shards: shard 0 device %.4f s wall %.4f s -> efficiency vs full/n: device %.3f wall %.3f" % (st.seconds, wall, st1.seconds / n / st.seconds, st1.seconds / n / wall))
