"""usage: fuzz_parity.py [n] [first_seed] [small|shards] — GPU vs oracle on n synthetic scenes (tests/oracle.py random_scene) with random
image sizes, depths, sample counts, generators and estimators. `small`: only scenes that take k_path's instantiation without the
cold features (<= 24 surfaces, no media, motion or textures, reference estimator: the LDS walk with its med3 / 64-bit-key tests),
with sample counts that span several summation blocks. `shards`: the general scenes rendered as what one rank of an N-GPU job or a
progressive pass renders - a random row range, an interleaved row stride of 1..8, a random number of samples kept in flight per
pass, larger sample counts. Prints every mismatch and a summary; exit code 1 on any."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from raytracing_weekend_amd import abi
import oracle

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
first = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
small = len(sys.argv) > 3 and sys.argv[3] == "small"
shards = len(sys.argv) > 3 and sys.argv[3] == "shards"
r = abi.Renderer(0)
bad = 0
for seed in range(first, first + n):
    rs = np.random.RandomState(seed)
    kw = dict(n_prims=int(rs.choice([4, 9, 16, 24, 25, 40, 90, 200])), volumes=bool(rs.randint(2)), motion=bool(rs.randint(2)),
              n_lights=int(rs.randint(0, 4)), sky=bool(rs.randint(2)), textured=bool(rs.randint(3) == 0))
    if small:
        kw.update(n_prims=int(rs.randint(2, 25)), volumes=False, motion=False, textured=False)
    w, h = int(rs.randint(9, 120)), int(rs.randint(9, 90))
    blob = oracle.random_scene(seed, w, h, **kw)
    p = abi.make_params(w, h, int(rs.randint(1, 9)), int(rs.choice([1, 2, 5, 12, 50])), rng_kind=int(rs.randint(2)),
                        seed=int(rs.randint(1, 1 << 31)), estimator=int(rs.choice([0, 0, 1, 2, 3])), sample_offset=int(rs.choice([0, 0, 7])))
    if small:
        p.estimator = 0
        p.spp = int(rs.choice([1, 3, 63, 64, 65, 130, 200]))
    if shards:
        p.spp = int(rs.choice([1, 5, 17, 40, 70]))
        p.row0 = int(rs.randint(0, h))
        p.row1 = int(rs.randint(p.row0 + 1, h + 1))
        p.row_stride = int(rs.randint(0, 9))
        p.samples_per_pass = int(rs.choice([0, 0, 1, 3, 16, 33]))
    r.upload_scene(blob)
    img, st = r.render(p)
    ref, st_ref = oracle.render(blob, p, threads=16)
    ok = np.array_equal(img, ref) and st.segments == st_ref.segments and st.shadow_rays == st_ref.shadow_rays
    if not ok:
        bad += 1
        d = np.argwhere(np.any(img != ref, axis=-1))
        print("MISMATCH seed", seed, kw, (w, h), dict(spp=p.spp, depth=p.max_depth, rng=p.rng_kind, est=p.estimator),
              "pixels", len(d), "first", d[:3].tolist(), "seg", st.segments, st_ref.segments, flush=True)
    if (seed - first) % 50 == 49:
        print("checked", seed - first + 1, "mismatches", bad, flush=True)
print("done:", n, "scenes,", bad, "mismatches")
sys.exit(1 if bad else 0)
