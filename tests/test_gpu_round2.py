"""GPU parity tests added in round 2 (-m gpu), all through the C ABI and against the CPU oracle bit for bit:
inputs the round-1 suite never ran: BASELINE config 5's dimensions, the headline workload at its real
sample range under the default schedule (both pipelines), a thin-lens camera, a light sampled through RTW_PDF_RECT_X,
trees whose root is a leaf, and the in-library multi-device render (rtw_create with n_devices > 1)."""
import ctypes as C

import numpy as np
import pytest

import oracle
from raytracing_weekend_amd import abi

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    r = abi.Renderer(0)
    yield r
    r.close()


def check(img, ref, st=None, st_ref=None):
    assert np.isfinite(img).all()
    d = img[..., :3].astype(np.float64) - ref[..., :3]
    assert float(np.sqrt(np.mean(d * d))) < 1e-4  # the north star's tolerance
    assert np.array_equal(img[..., :3], ref[..., :3])  # and the arithmetic spec's: exact
    assert np.all(img[..., 3] == 1.0)
    if st is not None and st_ref is not None:
        assert (st.samples, st.segments, st.shadow_rays) == (st_ref.samples, st_ref.segments, st_ref.shadow_rays)


@pytest.mark.parametrize("path", ["1", "0"])
def test_config5_dimensions_7680x4320(gpu, monkeypatch, path):
    """BASELINE config 5 (Cornell box 7680x4320): the last rows of the frame (pixel ids up to 33 M) and the interleaved
    shard rank 7 of 8 renders, 2 spp at depth 50, both pipelines."""
    monkeypatch.setenv("RTW_PATH", path)
    W, H = 7680, 4320
    blob = abi.build_scene(0, W, H)
    gpu.upload_scene(blob)
    for kw in (dict(row0=4316, row1=4320), dict(row0=7, row1=H, row_stride=8)):
        p = abi.make_params(W, H, 2, 50, **kw)
        img, st = gpu.render(p)
        ref, st_ref = oracle.render(blob, p, threads=32)
        assert img.shape[0] == abi.local_rows(p)
        check(img, ref, st, st_ref)


@pytest.mark.parametrize("path", ["1", "0"])
def test_headline_workload_band_at_4096_spp(gpu, monkeypatch, path):
    """The metric workload itself (Cornell box 1920x1080, 4096 spp, depth 50, default pool / lanes / stagger / job size):
    the full frame is rendered exactly as bench.py renders it, and rows 538-541 are compared with the oracle, which
    covers sample indices up to 4095, all 64 sample blocks and - for the wavefront pipeline - the 65-batch, two-lane,
    staggered schedule."""
    monkeypatch.setenv("RTW_PATH", path)
    W, H, spp = 1920, 1080, 4096
    blob = abi.build_scene(0, W, H)
    gpu.upload_scene(blob)
    full, st = gpu.render(abi.make_params(W, H, spp, 50))
    assert st.samples == W * H * spp
    p = abi.make_params(W, H, spp, 50, row0=538, row1=542)
    ref, _ = oracle.render(blob, p, threads=64)
    check(full[538:542], ref)
    band, st_b = gpu.render(p)  # the same rows as a shard of their own
    assert np.array_equal(band, full[538:542])


def test_thin_lens_camera(gpu):
    """lens_radius != 0 (camera.cu:11-19, sampling.cuh:15-22): the reference never sets it, the blob may."""
    w, h = 96, 72
    for scene in (0, 1):
        parts = dict(abi.parse_scene(abi.build_scene(scene, w, h)))
        hdr = abi.SceneHeader.from_buffer_copy(bytes(parts["header"]))
        hdr.camera.lens_radius = 0.5 if scene == 0 else 0.05
        parts["header"] = hdr
        blob = abi.assemble_scene(parts)
        gpu.upload_scene(blob)
        for rng in (abi.RTW_RNG_PHILOX, abi.RTW_RNG_TEA_LCG):
            p = abi.make_params(w, h, 5, 20, rng_kind=rng)
            img, st = gpu.render(p)
            ref, st_ref = oracle.render(blob, p, threads=16)
            check(img, ref, st, st_ref)
        sharp, _ = gpu.render(abi.make_params(w, h, 5, 20))
        gpu.upload_scene(abi.build_scene(scene, w, h))
        pin, _ = gpu.render(abi.make_params(w, h, 5, 20))
        assert not np.array_equal(sharp, pin)  # the lens does something


def x_rect_light_scene(w, h, extra=0):
    """Cornell box whose sampled light is an x-rectangle on the x = 554 wall (pdf.p1_gen = RTW_PDF_RECT_X, rectPdf.cu:124-146)."""
    parts = dict(abi.parse_scene(abi.build_scene(0, w, h) if extra == 0 else oracle.cluttered_cornell(w, h, n_extra=extra)))
    mats = list(parts["materials"])
    i_light = next(i for i, m in enumerate(mats) if m.type == abi.MAT_DIFFUSE_LIGHT)
    prims = list(parts["prims"])
    pr = abi.Prim(type=abi.PRIM_RECT_X, material=i_light, xform=0, flip=1)
    for k, v in enumerate((200.0, 350.0, 180.0, 330.0, 554.0)):
        pr.p[k] = v
    prims.append(pr)
    lt = abi.Light()
    lt.position[0], lt.position[1], lt.position[2] = 554.0, 200.0, 180.0
    lt.vec_u[1], lt.vec_v[2] = 150.0, 150.0
    lt.normal[0] = -1.0
    lt.area = 22500.0
    lt.emission[0], lt.emission[1], lt.emission[2] = 15.0, 15.0, 15.0
    hdr = abi.SceneHeader.from_buffer_copy(bytes(parts["header"]))
    hdr.pdf.p1_gen = abi.RTW_PDF_RECT_X
    for k, v in enumerate((200.0, 350.0, 180.0, 330.0, 554.0)):
        hdr.pdf.rect[k] = v
    parts.update(header=hdr, prims=prims, lights=[lt])
    return abi.assemble_scene(parts)


@pytest.mark.parametrize("extra", [0, 30])
def test_light_sampled_through_rect_x_pdf(gpu, extra):
    w, h = 96, 72
    blob = x_rect_light_scene(w, h, extra)
    gpu.upload_scene(blob)
    for rng in (abi.RTW_RNG_PHILOX, abi.RTW_RNG_TEA_LCG):
        p = abi.make_params(w, h, 6, 30, rng_kind=rng)
        img, st = gpu.render(p)
        ref, st_ref = oracle.render(blob, p, threads=16)
        assert st_ref.shadow_rays > 0
        check(img, ref, st, st_ref)


def tiny_scene(n_rects, n_volumes, w, h):
    """Sky-lit scene of n_rects surfaces (+ n_volumes fog spheres): with more than 24 primitives, or RTW_BRUTE_MAX=0, the
    library builds a tree whose root is a leaf when n_rects <= 2."""
    parts = dict(abi.parse_scene(abi.build_scene(3, w, h)))  # scene 3 carries an isotropic material for the media
    hdr = abi.SceneHeader.from_buffer_copy(bytes(parts["header"]))
    hdr.sky_light = 1
    mats = list(parts["materials"])
    i_lam = next(i for i, m in enumerate(mats) if m.type == abi.MAT_LAMBERTIAN)
    i_iso = next(i for i, m in enumerate(mats) if m.type == abi.MAT_ISOTROPIC)
    prims = []
    rects = [(abi.PRIM_RECT_Y, (0.0, 555.0, 0.0, 555.0, 0.0)), (abi.PRIM_RECT_Z, (100.0, 455.0, 50.0, 400.0, 300.0))]
    for t, q in rects[:n_rects]:
        pr = abi.Prim(type=t, material=i_lam, xform=0, flip=0)
        for k, v in enumerate(q):
            pr.p[k] = v
        prims.append(pr)
    rs = np.random.RandomState(5)
    for _ in range(n_volumes):
        pr = abi.Prim(type=abi.PRIM_VOLUME_SPHERE, material=i_iso, xform=0, flip=0)
        c = rs.uniform(80, 470, 3)
        for k, v in enumerate((c[0], c[1], c[2], rs.uniform(20, 60), 0.01)):
            pr.p[k] = float(v)
        prims.append(pr)
    parts.update(header=hdr, prims=prims, lights=[])
    return abi.assemble_scene(parts)


@pytest.mark.parametrize("n_rects,n_volumes,brute_max", [(1, 0, "0"), (2, 0, "0"), (2, 25, None), (1, 30, None), (0, 26, None)])
def test_tree_whose_root_is_a_leaf(gpu, monkeypatch, n_rects, n_volumes, brute_max):
    """one or two surfaces under a forced tree, and a > 24-primitive scene that is nearly all fog: the
    surfaces used to vanish (and an axis-aligned ray could loop). Rays through the origin along the axes are included."""
    if brute_max is not None:
        monkeypatch.setenv("RTW_BRUTE_MAX", brute_max)
    w, h = 64, 48
    blob = tiny_scene(n_rects, n_volumes, w, h)
    gpu.upload_scene(blob)
    for rng in (abi.RTW_RNG_PHILOX, abi.RTW_RNG_TEA_LCG):
        p = abi.make_params(w, h, 4, 12, rng_kind=rng)
        img, st = gpu.render(p)
        ref, st_ref = oracle.render(blob, p, threads=8)
        check(img, ref, st, st_ref)
    if n_rects:
        rays = np.array([[0, 100, 0, 0, -1, 0, 1e-6, 1e27], [278, 278, -800, 0, 0, 1, 1e-6, 1e27], [0, 0, 0, 0, -1, 0, 1e-6, 1e27],
                         [0, 0, 0, 1, 0, 0, 1e-6, 1e27], [300, 500, 100, 0, -1, 0, 1e-6, 1e27]], dtype=np.float32)
        t, prim = gpu.debug_intersect(rays)
        t_ref, prim_ref = oracle.intersect(blob, rays)
        assert np.array_equal(prim, prim_ref) and np.array_equal(t, t_ref)
    monkeypatch.delenv("RTW_BRUTE_MAX", raising=False)
    gpu.upload_scene(abi.build_scene(0, 16, 16))


@pytest.mark.parametrize("scene", [100, 200, 101, 204])
def test_environment_and_orthographic_cameras(gpu, scene):
    """the reference's two unused camera kinds (scene/ioCamera.h:118-179, scene/camera.cuh:35-56) through
    the C++ host description (scene + 100 * kind) and the blob's camera_type, GPU vs oracle, both generators, both pipelines'
    raygen (k_path on the small scenes, k_first on the trees)."""
    w, h = 96, 64
    blob = abi.build_scene(scene, w, h)
    assert abi.parse_scene(blob)["header"].camera_type == scene // 100
    gpu.upload_scene(blob)
    for rng in (abi.RTW_RNG_PHILOX, abi.RTW_RNG_TEA_LCG):
        p = abi.make_params(w, h, 4, 16, rng_kind=rng)
        img, st = gpu.render(p)
        ref, st_ref = oracle.render(blob, p, threads=16)
        check(img, ref, st, st_ref)
    if scene != 200:  # (the small orthographic window of the Cornell box looks straight at the mirror box, which shows the dark outside)
        assert img[..., :3].max() > 0
    pers, _ = oracle.render(abi.build_scene(scene % 100, w, h), abi.make_params(w, h, 4, 16, rng_kind=rng), threads=16)
    assert not np.array_equal(pers, ref)


@pytest.mark.parametrize("ids", [[0, 0], [0, 0, 0]])
def test_in_library_multi_device_render_is_the_single_device_image(gpu, ids):
    """rtw_create(n_devices = N): N interleaved shards rendered from N host threads, gathered on device_ids[0]; with every
    entry naming GPU 0 the frame must equal the single-context render bit for bit (Director.cpp:971-1008 sits on this)."""
    grp = abi.Renderer(ids)
    try:
        for scene, w, h, spp, depth in ((0, 200, 131, 70, 30), (1, 96, 64, 4, 12)):
            blob = abi.build_scene(scene, w, h)
            gpu.upload_scene(blob)
            grp.upload_scene(blob)
            for kw in (dict(), dict(row0=3, row1=h - 2), dict(row0=1, row1=h, row_stride=2)):
                p = abi.make_params(w, h, spp, depth, **kw)
                one, st1 = gpu.render(p)
                many, stn = grp.render(p)
                assert np.array_equal(one, many)
                assert (st1.samples, st1.segments, st1.shadow_rays) == (stn.samples, stn.segments, stn.shadow_rays)
    finally:
        grp.close()


def test_group_context_errors():
    lib = abi.load_hip()
    ctx = C.c_void_p()
    assert lib.rtw_create(C.byref(ctx), 0, None) == -1
    ids = (C.c_int * 2)(0, 0)
    assert lib.rtw_create(C.byref(ctx), 2, ids) == 0
    p = abi.make_params(8, 8, 1, 2)
    out = np.empty((8, 8, 4), np.float32)
    assert lib.rtw_render(ctx, C.byref(p), out.ctypes.data, None) == -3  # no scene yet
    assert lib.rtw_destroy(ctx) == 0


@pytest.mark.parametrize("env", [{}, {"RTW_LANES": "1", "RTW_TAIL_START": "2"}])
def test_tree_beyond_16_bit_references(gpu, monkeypatch, env):
    """9 000 primitives (moving spheres among them, lights, sky): the leaf table has more slots than a 16-bit stack entry
    can name, so the walk takes its 32-bit form (k_trace_bvh mode 0, the step with branches) - a form no reference scene
    reaches. Wavefront kernels, and the fused tail from depth 2 (traverse<>)."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    W, H = 48, 36
    blob = oracle.random_scene(77, W, H, n_prims=9000, motion=True, n_lights=2, sky=True)
    gpu.upload_scene(blob)
    p = abi.make_params(W, H, 2, 8)
    img, st = gpu.render(p)
    ref, st_ref = oracle.render(blob, p, threads=16)
    check(img, ref, st, st_ref)


@pytest.mark.parametrize("lds_kb", ["6", "60"])
def test_tree_image_in_lds_or_global(gpu, monkeypatch, lds_kb):
    """The same 300-primitive tree with almost none of it in the trace kernel's LDS image (mode 1: nodes and leaf slots
    from global memory) and with all of it (mode 2), moving spheres in their two-slot form."""
    monkeypatch.setenv("RTW_TRACE_LDS_KB", lds_kb)
    W, H = 64, 48
    blob = oracle.random_scene(78, W, H, n_prims=300, motion=True, n_lights=1)
    gpu.upload_scene(blob)
    p = abi.make_params(W, H, 3, 12)
    img, st = gpu.render(p)
    ref, st_ref = oracle.render(blob, p, threads=16)
    check(img, ref, st, st_ref)
