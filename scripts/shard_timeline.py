"""usage: shard_timeline.py [n] [spp] — renders the 1/n interleaved shard of the metric frame twice (warm-up, then the render whose
launches are to be read from a rocprofv3 --kernel-trace of this script) and prints the device seconds of both."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracing_weekend_amd import abi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
W, H, D = 1920, 1080, 50
r = abi.Renderer(0); r.upload_scene(abi.build_scene(0, W, H))
p = abi.make_params(W, H, spp, D, row0=0, row1=H, row_stride=n)
for _ in range(2):
    _, st = r.render(p)
    print("device seconds", round(st.seconds, 5), flush=True)
