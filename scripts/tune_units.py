"""Device seconds of the full metric frame and of its 1/8 interleaved shard for a few (unit blocks, fine blocks) settings
(RTW_PATH_UNIT_BLOCKS, RTW_PATH_FINE_BLOCKS are read per render call)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracing_weekend_amd import abi
W, H, D, SPP = 1920, 1080, 50, 4096
r = abi.Renderer(0); r.upload_scene(abi.build_scene(0, W, H))
for U, F in ((4, 8), (4, 16), (4, 4), (8, 8), (2, 8), (4, 0), (16, 16)):
    os.environ["RTW_PATH_UNIT_BLOCKS"], os.environ["RTW_PATH_FINE_BLOCKS"] = str(U), str(F)
    out = []
    for kw in (dict(), dict(row0=0, row1=H, row_stride=8)):
        p = abi.make_params(W, H, SPP, D, **kw)
        r.render(p)
        out.append(min(r.render(p)[1].seconds for _ in range(2)))
    print("U=%d F=%d full %.4f s  shard %.4f s  shard efficiency %.3f" % (U, F, out[0], out[1], out[0] / 8 / out[1]), flush=True)
