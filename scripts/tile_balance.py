"""Times the 8 row tiles of the metric frame separately on one GPU: predicts the strong-scaling balance of the 8-GPU split."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracing_weekend_amd import abi
W, H, SPP, D = 1920, 1080, 1024, 50
r = abi.Renderer(0); r.upload_scene(abi.build_scene(0, W, H))
for n in (2, 4, 8):
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
