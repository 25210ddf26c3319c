"""Experiment: cost of material divergence in k_path. Renders the Cornell box as is, and with the metal box / glass sphere
turned into Lambertian surfaces (same geometry), and prints Msamples/s, segments per sample and Gsegments/s."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracing_weekend_amd import abi
w, h, spp, depth = 1920, 1080, int(sys.argv[1]) if len(sys.argv) > 1 else 512, 50
r = abi.Renderer(0)
def run(tag, blob):
    r.upload_scene(blob)
    r.render(abi.make_params(w, h, 8, depth))
    _, st = r.render(abi.make_params(w, h, spp, depth))
    print(json.dumps({"case": tag, "Msamples_per_s": round(st.samples / st.seconds / 1e6, 1), "seg_per_sample": round(st.segments / st.samples, 3),
                      "shadow_per_sample": round(st.shadow_rays / st.samples, 3), "Gseg_per_s": round(st.segments / st.seconds / 1e9, 2)}))
blob = abi.build_scene(0, w, h)
run("cornell", blob)
parts = dict(abi.parse_scene(blob))
mats = list(parts["materials"])
lam = next(m for m in mats if m.type == abi.MAT_LAMBERTIAN)
for kinds, tag in (((abi.MAT_METAL,), "metal->lambertian"), ((abi.MAT_METAL, abi.MAT_DIELECTRIC), "metal,glass->lambertian")):
    m2 = []
    for m in mats:
        if m.type in kinds:
            m2.append(abi.Material(type=abi.MAT_LAMBERTIAN, texture=lam.texture, fuzz_or_eta=0.0, bsdf_eval=0))
        else:
            m2.append(m)
    parts["materials"] = m2
    run(tag, abi.assemble_scene(parts))
