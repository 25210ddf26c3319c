"""Device seconds of the full metric frame and of its 1/8 interleaved shard for a few (unit blocks, fine blocks, units per job)
settings (RTW_PATH_UNIT_BLOCKS, RTW_PATH_FINE_BLOCKS, RTW_PATH_JOB_BLOCKS are read per render call; 0 unit blocks = automatic)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracing_weekend_amd import abi
W, H, D, SPP = 1920, 1080, 50, 4096
r = abi.Renderer(0); r.upload_scene(abi.build_scene(0, W, H))
CASES = [(0, 8, 2), (4, 8, 2), (8, 8, 2), (8, 8, 1), (8, 8, 4), (8, 16, 2), (8, 24, 2), (16, 16, 2), (4, 16, 2), (2, 8, 2)]
for U, F, J in CASES:
    os.environ["RTW_PATH_UNIT_BLOCKS"], os.environ["RTW_PATH_FINE_BLOCKS"], os.environ["RTW_PATH_JOB_BLOCKS"] = str(U), str(F), str(J)
    out = []
    for kw in (dict(), dict(row0=0, row1=H, row_stride=8)):
        p = abi.make_params(W, H, SPP, D, **kw)
        r.render(p)
        out.append(min(r.render(p)[1].seconds for _ in range(2)))
    print("U=%d F=%d J=%d full %.4f s  1/8 shard %.4f s" % (U, F, J, out[0], out[1]), flush=True)
