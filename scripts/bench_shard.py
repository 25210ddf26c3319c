"""usage: bench_shard.py <n> <spp> [reps] - device seconds of interleaved shard 0 of n of the metric frame (and of the full frame):
what one rank of an n-GPU run renders. Prints one JSON line."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracing_weekend_amd import abi
n, spp = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
W, H, D = 1920, 1080, 50
r = abi.Renderer(0); r.upload_scene(abi.build_scene(0, W, H))
p = abi.make_params(W, H, spp, D, row0=0, row1=H, row_stride=n)
r.render(abi.make_params(W, H, 8, D, row0=0, row1=H, row_stride=n))
ts = [r.render(p)[1].seconds for _ in range(reps)]
print(json.dumps({"lib": os.path.basename(abi.HIP_LIB), "shards": n, "spp": spp, "seconds": [round(t, 5) for t in ts], "best": round(min(ts), 5)}))
