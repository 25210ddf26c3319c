"""Loader for the CPU oracle (oracle/librtw_oracle.so). TEST INFRASTRUCTURE: imported only by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg — never by the product package."""
import ctypes as C
import os
import subprocess

import numpy as np

from raytracing_weekend_amd import abi

ORACLE_DIR = os.path.join(abi.REPO_DIR, "oracle")
ORACLE_LIB = os.path.join(ORACLE_DIR, "librtw_oracle.so")
REF_RANDOM_LIB = os.path.join(ORACLE_DIR, "_ref", "libref_random.so")
_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(ORACLE_LIB):
            subprocess.check_call(["make", "-C", ORACLE_DIR, "librtw_oracle.so"])
        lib = C.CDLL(ORACLE_LIB)
        lib.rtwo_render.restype = C.c_int
        lib.rtwo_render.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(abi.Params), C.c_void_p, C.POINTER(abi.Stats), C.c_int]
        lib.rtwo_intersect.restype = C.c_int
        lib.rtwo_intersect.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        lib.rtwo_trace_pixel.restype = C.c_int
        lib.rtwo_trace_pixel.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(abi.Params), C.c_int, C.c_int, C.c_int, C.c_void_p]
        lib.rtwo_tea.restype = C.c_uint32
        lib.rtwo_tea.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32]
        lib.rtwo_xorshift32.restype = C.c_uint32
        lib.rtwo_xorshift32.argtypes = [C.POINTER(C.c_uint32)]
        lib.rtwo_randf.restype = C.c_float
        lib.rtwo_randf.argtypes = [C.POINTER(C.c_uint32)]
        lib.rtwo_lcg_rnd.restype = C.c_float
        lib.rtwo_lcg_rnd.argtypes = [C.POINTER(C.c_uint32)]
        lib.rtwo_philox4x32_10.restype = None
        lib.rtwo_philox4x32_10.argtypes = [C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        lib.rtwo_logf.restype = C.c_float
        lib.rtwo_logf.argtypes = [C.c_float]
        lib.rtwo_sincos2pi.restype = None
        lib.rtwo_sincos2pi.argtypes = [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        _lib = lib
    return _lib


def render(blob, params, threads=1):
    lib = load()
    rows = abi.local_rows(params)
    out = np.empty((rows, params.width, 4), dtype=np.float32)
    st = abi.Stats()
    rc = lib.rtwo_render(blob, len(blob), C.byref(params), out.ctypes.data, C.byref(st), threads)
    if rc != 0:
        raise RuntimeError(f"rtwo_render failed: {rc}")
    return out, st


def intersect(blob, rays, ray_time=None, gather_time=None):
    lib = load()
    rays = np.ascontiguousarray(rays, dtype=np.float32)
    n = rays.shape[0]
    rt = None if ray_time is None else np.ascontiguousarray(ray_time, dtype=np.float32)
    gt = None if gather_time is None else np.ascontiguousarray(gather_time, dtype=np.float32)
    t = np.empty(n, dtype=np.float32)
    prim = np.empty(n, dtype=np.int32)
    rc = lib.rtwo_intersect(blob, len(blob), rays.ctypes.data, None if rt is None else rt.ctypes.data,
                            None if gt is None else gt.ctypes.data, n, t.ctypes.data, prim.ctypes.data)
    if rc != 0:
        raise RuntimeError(f"rtwo_intersect failed: {rc}")
    return t, prim


def cluttered_cornell(w, h, n_extra=40, seed=7, scene=0):
    """Cornell box (reference scene 0) with n_extra small spheres added: > 24 primitives, so the HIP library walks
    its BVH, and the scene keeps its rect light, so queued shadow probes go through the tree as well. Synthetic
    (not a reference scene): exercises tree + light sampling + transforms together."""
    import numpy as np
    from raytracing_weekend_amd import abi
    parts = abi.parse_scene(abi.build_scene(scene, w, h))  # scene=3: the fog boxes stay in, so the fused path walks the tree
    prims = list(parts["prims"])
    n_mat = len(parts["materials"])
    rs = np.random.RandomState(seed)
    for i in range(n_extra):
        pr = abi.Prim(type=abi.PRIM_SPHERE, material=int(rs.randint(0, n_mat)), xform=0, flip=0)
        r = float(rs.uniform(12.0, 30.0))
        pr.p[0], pr.p[1], pr.p[2], pr.p[3] = (float(rs.uniform(60, 495)), float(rs.uniform(40, 500)), float(rs.uniform(60, 495)), r)
        prims.append(pr)
    parts = dict(parts)
    parts["prims"] = prims
    return abi.assemble_scene(parts)


def random_scene(seed, w, h, n_prims=30, volumes=False, motion=False, n_lights=1, sky=False, textured=False):
    """A synthetic scene for fuzzing GPU-vs-oracle parity: every primitive kind under random rigid transforms, every
    material kind, 0..n rectangle lights in the light list, optional sky light / volumes / moving spheres. The camera,
    the image plane and the pdf record come from the Cornell box header (the content sits in its 0..555 cube)."""
    import ctypes as C
    import numpy as np
    from raytracing_weekend_amd import abi
    rs = np.random.RandomState(seed)
    parts = dict(abi.parse_scene(abi.build_scene(0, w, h)))
    hdr = abi.SceneHeader.from_buffer_copy(bytes(parts["header"]))
    hdr.sky_light = 1 if sky else 0

    def xform(rot_axis, deg, t):
        a = np.deg2rad(deg)
        c, s = np.cos(a), np.sin(a)
        r = np.eye(4)
        i, j = [(1, 2), (0, 2), (0, 1)][rot_axis]
        r[i, i], r[i, j], r[j, i], r[j, j] = c, -s, s, c
        m = np.eye(4)
        m[:3, 3] = t
        m = m @ r
        x = abi.Xform()
        mi = np.linalg.inv(m)
        for k in range(12):
            x.m[k] = float(np.float32(m[k // 4, k % 4]))
            x.inv[k] = float(np.float32(mi[k // 4, k % 4]))
        return x
    ident = abi.Xform()
    for k in (0, 5, 10):
        ident.m[k] = ident.inv[k] = 1.0
    xforms = [ident] + [xform(int(rs.randint(3)), float(rs.uniform(-40, 40)), rs.uniform(-60, 60, 3)) for _ in range(3)]

    textures, materials = [], []

    def tex(rgb):
        t = abi.Texture(type=abi.TEX_CONSTANT)
        t.color[0], t.color[1], t.color[2] = (float(v) for v in rgb)
        textures.append(t)
        return len(textures) - 1
    for _ in range(3):
        materials.append(abi.Material(type=abi.MAT_LAMBERTIAN, texture=tex(rs.uniform(0.1, 0.9, 3)), fuzz_or_eta=0.0, bsdf_eval=0))
    materials.append(abi.Material(type=abi.MAT_METAL, texture=tex(rs.uniform(0.5, 0.95, 3)), fuzz_or_eta=float(rs.uniform(0, 0.5)), bsdf_eval=2))
    materials.append(abi.Material(type=abi.MAT_DIELECTRIC, texture=-1, fuzz_or_eta=1.5, bsdf_eval=1))
    i_light = len(materials)
    materials.append(abi.Material(type=abi.MAT_DIFFUSE_LIGHT, texture=tex((9.0, 8.0, 7.0)), fuzz_or_eta=0.0, bsdf_eval=-1))
    materials.append(abi.Material(type=abi.MAT_NORMAL, texture=-1, fuzz_or_eta=0.0, bsdf_eval=-1))
    i_iso = len(materials)
    materials.append(abi.Material(type=abi.MAT_ISOTROPIC, texture=tex(rs.uniform(0.3, 0.9, 3)), fuzz_or_eta=0.0, bsdf_eval=-1))
    surface = [0, 1, 2, 3, 4, 6]

    prims, lights = [], []

    def add(ptype, params, material, xf=0, flip=0):
        pr = abi.Prim(type=ptype, material=material, xform=xf, flip=flip)
        for k, v in enumerate(params):
            pr.p[k] = float(np.float32(v))
        prims.append(pr)
    # the light(s): y-rectangles near the ceiling, the first one is the rectangle the pdf record samples
    rect = [float(hdr.pdf.rect[k]) for k in range(5)]
    for li in range(n_lights):
        x0, x1, z0, z1, k = rect if li == 0 else (float(rs.uniform(50, 250)), float(rs.uniform(300, 500)), float(rs.uniform(50, 250)), float(rs.uniform(300, 500)), 540.0 - 10.0 * li)
        add(abi.PRIM_RECT_Y, (x0, x1, z0, z1, k), i_light, flip=1)
        lt = abi.Light()
        lt.position[0], lt.position[1], lt.position[2] = x0, k, z0
        lt.vec_u[0], lt.vec_u[1], lt.vec_u[2] = x1 - x0, 0.0, 0.0
        lt.vec_v[0], lt.vec_v[1], lt.vec_v[2] = 0.0, 0.0, z1 - z0
        lt.normal[0], lt.normal[1], lt.normal[2] = 0.0, -1.0, 0.0
        lt.area = float(np.float32((x1 - x0) * (z1 - z0)))
        lt.emission[0], lt.emission[1], lt.emission[2] = 9.0, 8.0, 7.0
        lights.append(lt)
    # a floor so that most paths bounce
    add(abi.PRIM_RECT_Y, (-200, 755, -200, 755, 0.0), 0)
    while len(prims) < n_prims:
        kind = rs.randint(0, 10)
        xf = int(rs.randint(0, len(xforms))) if rs.rand() < 0.5 else 0
        mat = surface[int(rs.randint(len(surface)))]
        c = rs.uniform(60, 495, 3)
        if kind < 4:
            add(abi.PRIM_SPHERE, (c[0], c[1], c[2], rs.uniform(15, 70)), mat, xf)
        elif kind < 5 and motion:
            c1 = c + rs.uniform(-40, 40, 3)
            add(abi.PRIM_MOVING_SPHERE, (c[0], c[1], c[2], rs.uniform(15, 50), c1[0], c1[1], c1[2], 0.0, 1.0), mat, xf)
        elif kind < 8:
            a0, b0 = rs.uniform(20, 400, 2)
            add(abi.PRIM_RECT_X + int(rs.randint(3)), (a0, a0 + rs.uniform(30, 200), b0, b0 + rs.uniform(30, 200), rs.uniform(20, 535)), mat, xf, int(rs.randint(2)))
        elif kind < 9 and volumes:
            lo = rs.uniform(60, 350, 3)
            add(abi.PRIM_VOLUME_BOX, (lo[0], lo[1], lo[2], lo[0] + rs.uniform(60, 180), lo[1] + rs.uniform(60, 180), lo[2] + rs.uniform(60, 180), rs.uniform(0.002, 0.02)), i_iso, xf)
        elif volumes:
            add(abi.PRIM_VOLUME_SPHERE, (c[0], c[1], c[2], rs.uniform(40, 110), rs.uniform(0.002, 0.02)), i_iso, xf)
    texdata = b""
    if textured:  # the three Lambertian materials and the metal get noise / checker / image / checker-of-those textures
        texdata = perlin_tables(seed) + test_image(seed=seed)
        t_noise = abi.Texture(type=abi.TEX_NOISE, scale=float(rs.choice([0.02, 0.1, 1.0])), data=0)
        t_img = abi.Texture(type=abi.TEX_IMAGE, data=1536)
        textures.extend([t_noise, t_img])
        i_noise, i_img = len(textures) - 2, len(textures) - 1
        textures.append(abi.Texture(type=abi.TEX_CHECKER, odd=0, even=i_img))
        textures.append(abi.Texture(type=abi.TEX_CHECKER, odd=i_noise, even=1))
        for m, t in zip(range(4), (i_noise, i_img, len(textures) - 2, len(textures) - 1)):
            materials[m].texture = t
    parts.update(header=hdr, prims=prims, xforms=xforms, materials=materials, textures=textures, lights=lights, texdata=texdata)
    return abi.assemble_scene(parts)


def perlin_tables(seed):
    """ranvec[256][3] (unit vectors) + three permutations of 0..255, as the 1536 words rtw_texture.data points at."""
    import numpy as np
    rs = np.random.RandomState(seed)
    v = rs.uniform(-1, 1, (256, 3))
    v = (v / np.linalg.norm(v, axis=1, keepdims=True)).astype("<f4")
    perms = [rs.permutation(256).astype("<i4") for _ in range(3)]
    return v.tobytes() + b"".join(p.tobytes() for p in perms)


def test_image(w=64, h=32, seed=3):
    """A small synthetic RGBA8 image (smooth gradients + blocks), as the words rtw_texture.data points at."""
    import numpy as np
    rs = np.random.RandomState(seed)
    y, x = np.mgrid[0:h, 0:w]
    img = np.zeros((h, w, 4), np.uint8)
    img[..., 0] = (255 * x / (w - 1)).astype(np.uint8)
    img[..., 1] = (255 * y / (h - 1)).astype(np.uint8)
    img[..., 2] = rs.randint(0, 256, (h // 4, w // 4)).repeat(4, 0).repeat(4, 1)
    img[..., 3] = 255
    return np.array([w, h], "<u4").tobytes() + img.astype("<u1").tobytes()


def textured_cornell(w, h, extra=0):
    """Cornell box (reference scene 0) whose materials get the non-constant textures: Perlin noise on the floor
    material, a checker of two constants on the back wall's, an image on the glass sphere's replacement (a Lambertian
    sphere) and on the light-less ceiling rectangle, a checker of (noise, image) on the tall box. extra > 0 adds
    spheres so that the tree path is used. Synthetic: exercises every texture callable and both u,v sources."""
    import numpy as np
    from raytracing_weekend_amd import abi
    parts = dict(abi.parse_scene(abi.build_scene(0, w, h)))
    prims, mats, texs = list(parts["prims"]), list(parts["materials"]), list(parts["textures"])
    texdata = bytearray()

    def blob_words(b):
        off = len(texdata) // 4
        texdata.extend(b)
        return off

    def add_tex(**kw):
        t = abi.Texture(**{k: v for k, v in kw.items() if k != "color"})
        if "color" in kw:
            t.color[0], t.color[1], t.color[2] = kw["color"]
        texs.append(t)
        return len(texs) - 1
    t_noise = add_tex(type=abi.TEX_NOISE, scale=0.05, data=blob_words(perlin_tables(1)))
    t_noise2 = add_tex(type=abi.TEX_NOISE, scale=4.0, data=blob_words(perlin_tables(2)))
    t_img = add_tex(type=abi.TEX_IMAGE, data=blob_words(test_image()))
    t_a = add_tex(type=abi.TEX_CONSTANT, color=(0.9, 0.2, 0.1))
    t_b = add_tex(type=abi.TEX_CONSTANT, color=(0.1, 0.2, 0.9))
    t_chk = add_tex(type=abi.TEX_CHECKER, odd=t_a, even=t_b)
    t_chk2 = add_tex(type=abi.TEX_CHECKER, odd=t_noise2, even=t_img)

    def lambert(tex):
        mats.append(abi.Material(type=abi.MAT_LAMBERTIAN, texture=tex, fuzz_or_eta=0.0, bsdf_eval=0))
        return len(mats) - 1
    m_noise, m_img, m_chk, m_chk2 = lambert(t_noise), lambert(t_img), lambert(t_chk), lambert(t_chk2)
    mats.append(abi.Material(type=abi.MAT_METAL, texture=t_noise2, fuzz_or_eta=0.1, bsdf_eval=2))
    m_metal_noise = len(mats) - 1
    rects = [i for i, p in enumerate(prims) if abi.PRIM_RECT_X <= p.type <= abi.PRIM_RECT_Z and p.xform == 0 and mats[p.material].type == abi.MAT_LAMBERTIAN]
    boxes = [i for i, p in enumerate(prims) if p.xform != 0]
    spheres = [i for i, p in enumerate(prims) if p.type == abi.PRIM_SPHERE]
    for k, i in enumerate(rects):
        prims[i].material = (m_noise, m_chk, m_img, m_metal_noise, m_chk2)[k % 5]
    for k, i in enumerate(boxes):
        prims[i].material = (m_chk2, m_img, m_noise)[k % 3]
    for i in spheres:
        prims[i].material = m_img
    rs = np.random.RandomState(5)
    for k in range(extra):
        pr = abi.Prim(type=abi.PRIM_SPHERE, material=(m_img, m_noise, m_chk, m_chk2)[k % 4], xform=0, flip=0)
        pr.p[0], pr.p[1], pr.p[2], pr.p[3] = (float(rs.uniform(60, 495)), float(rs.uniform(40, 500)), float(rs.uniform(60, 495)), float(rs.uniform(12, 40)))
        prims.append(pr)
    parts.update(prims=prims, materials=mats, textures=texs, texdata=bytes(texdata))
    return abi.assemble_scene(parts)


def corrupted_scenes(w=24, h=16):
    """(name, blob) pairs every loader must reject (RTW_ERR_BAD_SCENE) - the oracle and the HIP library alike."""
    import ctypes as C
    import numpy as np
    from raytracing_weekend_amd import abi
    base = dict(abi.parse_scene(textured_cornell(w, h)))
    out = []

    def variant(name, edit):
        parts = {k: (list(v) if k not in ("header", "texdata") else v) for k, v in base.items()}
        parts["prims"] = [abi.Prim.from_buffer_copy(bytes(p)) for p in parts["prims"]]
        parts["materials"] = [abi.Material.from_buffer_copy(bytes(m)) for m in parts["materials"]]
        parts["textures"] = [abi.Texture.from_buffer_copy(bytes(t)) for t in parts["textures"]]
        parts["xforms"] = [abi.Xform.from_buffer_copy(bytes(x)) for x in parts["xforms"]]
        edit(parts)
        out.append((name, abi.assemble_scene(parts)))

    def set_attr(table, i, field, value):
        def edit(parts):
            setattr(parts[table][i], field, value)
        return edit

    def set_p(i, k, value):
        def edit(parts):
            parts["prims"][i].p[k] = value
        return edit
    n_mat, n_tex = len(base["materials"]), len(base["textures"])
    checker = [i for i, t in enumerate(base["textures"]) if t.type == abi.TEX_CHECKER][0]
    image = [i for i, t in enumerate(base["textures"]) if t.type == abi.TEX_IMAGE][0]
    variant("primitive type", set_attr("prims", 0, "type", 7))
    variant("primitive type negative", set_attr("prims", 1, "type", -1))
    variant("primitive xform", set_attr("prims", 2, "xform", 99))
    variant("primitive material", set_attr("prims", 3, "material", n_mat))
    variant("primitive material negative", set_attr("prims", 3, "material", -2))
    variant("primitive parameter nan", set_p(4, 1, float("nan")))
    variant("primitive parameter inf", set_p(5, 4, float("inf")))
    variant("material texture", set_attr("materials", 0, "texture", n_tex))
    variant("texture type", set_attr("textures", 0, "type", 5))
    variant("checker child", set_attr("textures", checker, "odd", n_tex))
    variant("checker child negative", set_attr("textures", checker, "even", -1))
    variant("checker nested", set_attr("textures", checker, "odd", checker))
    variant("image beyond the data section", set_attr("textures", image, "data", len(base["texdata"]) // 4 - 3))

    def nan_xform(parts):
        parts["xforms"][1].inv[3] = float("nan")
    variant("transform nan", nan_xform)

    def huge_image(parts):
        words = np.frombuffer(parts["texdata"], "<u4").copy()
        words[parts["textures"][image].data] = 30000
        parts["texdata"] = words.tobytes()
    variant("image size larger than its texels", huge_image)
    blob = abi.assemble_scene(base)
    hdr = abi.SceneHeader.from_buffer_copy(blob[:C.sizeof(abi.SceneHeader)])
    for name, field, value in (("prim table offset", "off_prims", len(blob)), ("texture data offset", "off_texdata", len(blob) + 4),
                               ("texture data misaligned", "off_texdata", hdr.off_texdata + 2), ("no transforms", "n_xforms", 0),
                               ("light table", "n_lights", 1 << 20)):
        h2 = abi.SceneHeader.from_buffer_copy(bytes(hdr))
        setattr(h2, field, value)
        out.append((name, bytes(h2) + blob[C.sizeof(abi.SceneHeader):]))
    return out


def denoise(img, iterations=5, sigma=0.5):
    """CPU restatement of rtw_denoise."""
    import ctypes as C
    import numpy as np
    lib = load()
    lib.rtwo_denoise.restype = C.c_int
    lib.rtwo_denoise.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float]
    src = np.ascontiguousarray(img, dtype=np.float32)
    out = np.empty_like(src)
    rc = lib.rtwo_denoise(src.ctypes.data, out.ctypes.data, src.shape[1], src.shape[0], iterations, sigma)
    if rc:
        raise RuntimeError(f"rtwo_denoise: {rc}")
    return out
