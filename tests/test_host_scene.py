"""Host scene description (raytracing_weekend_amd/host, librtw_host.so) against the constants the
reference's scene code states (files cited per assertion)."""
import collections
import ctypes as C
import math
import os

import numpy as np
import pytest

from raytracing_weekend_amd import abi

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_struct_sizes_match_header():
    assert C.sizeof(abi.Prim) == 64 and C.sizeof(abi.Xform) == 96 and C.sizeof(abi.Material) == 16
    assert C.sizeof(abi.Texture) == 32 and C.sizeof(abi.Light) == 64 and C.sizeof(abi.Pdf) == 48
    assert C.sizeof(abi.Camera) == 96 and C.sizeof(abi.Params) == 48 and C.sizeof(abi.Stats) == 160


def test_cornell_box_blob():
    s = abi.parse_scene(abi.build_scene(0, 800, 800))
    h = s["header"]
    assert (h.magic, h.version) == (abi.RTW_SCENE_MAGIC, abi.RTW_ABI_VERSION)
    assert (h.n_prims, h.n_materials, h.n_lights, h.sky_light, h.n_xforms) == (13, 13, 1, 0, 2)  # ioScene.h:491-627
    prims = s["prims"]
    assert prims[0].type == abi.PRIM_SPHERE and list(prims[0].p[:4]) == [190.0, 90.0, 190.0, 90.0]
    kinds = [p.type for p in prims]
    assert kinds[1:7] == [abi.PRIM_RECT_X, abi.PRIM_RECT_X, abi.PRIM_RECT_Y, abi.PRIM_RECT_Y, abi.PRIM_RECT_Z, abi.PRIM_RECT_Y]
    assert [p.flip for p in prims[1:7]] == [1, 0, 1, 0, 1, 1]
    assert list(prims[6].p[:5]) == [213.0, 343.0, 227.0, 332.0, pytest.approx(554.9)]  # light rect, ioScene.h:510
    # box = Z,Z,Y,Y,X,X with flip on the p0 faces (ioGeometryGroup.h:27-40), all under one transform
    assert kinds[7:13] == [abi.PRIM_RECT_Z, abi.PRIM_RECT_Z, abi.PRIM_RECT_Y, abi.PRIM_RECT_Y, abi.PRIM_RECT_X, abi.PRIM_RECT_X]
    assert [p.flip for p in prims[7:13]] == [1, 0, 1, 0, 1, 0]
    assert all(p.xform == 1 for p in prims[7:13]) and all(p.xform == 0 for p in prims[:7])
    assert [p.material for p in prims] == list(range(13))
    m = np.array(s["xforms"][1].m[:]).reshape(3, 4)
    c, sn = math.cos(math.radians(15)), math.sin(math.radians(15))
    assert np.allclose(m, [[c, 0, sn, 265], [0, 1, 0, 0], [-sn, 0, c, 295]], atol=1e-5)  # T(265,0,295)*R_y(15)
    inv = np.array(s["xforms"][1].inv[:]).reshape(3, 4)
    full = np.vstack([m, [0, 0, 0, 1]]) @ np.vstack([inv, [0, 0, 0, 1]])
    assert np.allclose(full, np.eye(4), atol=1e-4)
    mats = s["materials"]
    assert mats[0].type == abi.MAT_DIELECTRIC and mats[0].fuzz_or_eta == 1.5
    assert mats[6].type == abi.MAT_DIFFUSE_LIGHT
    assert mats[7].type == abi.MAT_METAL and mats[7].fuzz_or_eta == pytest.approx(0.018)
    tex = s["textures"]
    assert list(tex[mats[1].texture].color) == pytest.approx([0.12, 0.45, 0.15])  # green wall is instance 1
    assert list(tex[mats[2].texture].color) == pytest.approx([0.65, 0.05, 0.05])
    # light definition: cross((130,0,0),(0,0,105)) = (0,-13650,0)  (ioScene.h:605-611)
    lt = s["lights"][0]
    assert lt.area == 13650.0 and list(lt.normal) == [0.0, -1.0, 0.0] and list(lt.emission) == [15.0, 15.0, 15.0]
    assert (h.pdf.gen, h.pdf.p0_gen, h.pdf.p1_gen) == (2, 0, 4)  # mixture(cosine, rect_y), ioScene.h:107-112
    assert list(h.pdf.rect) == [213.0, 343.0, 227.0, 332.0, pytest.approx(554.9)]
    # camera: w=(0,0,-1), u=(-1,0,0), v=(0,1,0); half height tan(20 deg); lens radius never copied (Director.cpp:494-496)
    cam = h.camera
    assert list(cam.w) == [0.0, 0.0, -1.0] and list(cam.u) == [-1.0, 0.0, 0.0] and list(cam.v) == [0.0, 1.0, 0.0]
    assert cam.lower_left[1] == pytest.approx(278 - 10 * math.tan(math.radians(20)), abs=1e-4)
    assert cam.lower_left[2] == pytest.approx(-790.0)
    assert cam.vertical[1] == pytest.approx(20 * math.tan(math.radians(20)), abs=1e-4)
    assert (cam.lens_radius, cam.time0, cam.time1) == (0.0, 0.0, 1.0)


def test_moving_spheres_scene_counts():
    s = abi.parse_scene(abi.build_scene(1, 1920, 1080))
    h = s["header"]
    assert h.n_prims == 528 and h.n_materials == 528 and h.n_lights == 0 and h.sky_light == 1
    kinds = collections.Counter(p.type for p in s["prims"])
    assert kinds[abi.PRIM_MOVING_SPHERE] == 333 and kinds[abi.PRIM_SPHERE] == 195
    mt = collections.Counter(m.type for m in s["materials"])
    assert mt[abi.MAT_LAMBERTIAN] == 333 + 2 and mt[abi.MAT_METAL] == 71 + 1 and mt[abi.MAT_DIELECTRIC] == 36 + 2 * 42 + 1
    mv = [p for p in s["prims"] if p.type == abi.PRIM_MOVING_SPHERE][0]
    assert mv.p[5] == pytest.approx(mv.p[1] + 0.18) and mv.p[3] == pytest.approx(0.2) and (mv.p[7], mv.p[8]) == (0.0, 1.0)
    assert h.pdf.gen == 0  # cosine, ioScene.h:115


def test_volumes_scene():
    s = abi.parse_scene(abi.build_scene(3, 400, 400))
    h = s["header"]
    assert (h.n_prims, h.n_lights, h.sky_light, h.n_xforms) == (8, 0, 1, 3)  # no LightDefinition pushed
    prims = s["prims"]
    assert prims[6].type == abi.PRIM_VOLUME_BOX and prims[6].p[6] == pytest.approx(0.006)
    assert prims[7].type == abi.PRIM_VOLUME_SPHERE and list(prims[7].p[:5]) == pytest.approx([82.5, 75.0, 82.5, 75.0, 0.005])
    assert s["materials"][6].type == abi.MAT_ISOTROPIC and s["materials"][7].type == abi.MAT_ISOTROPIC
    assert list(prims[5].p[:5]) == [213.0, 343.0, 227.0, 332.0, 554.0]
    t2 = np.array(s["xforms"][prims[7].xform].m[:]).reshape(3, 4)
    assert np.allclose(t2, [[1, 0, 0, 130], [0, 1, 0, 0], [0, 0, 1, 65]])


def test_unknown_scene_is_rejected():
    for bad in (2, 4, 5, -1):  # 2 and 4 need checker/noise/image textures: out of scope this round
        with pytest.raises(ValueError):
            abi.build_scene(bad, 64, 64)


def test_host_builder_reproduces_fixture_blobs():
    for name in os.listdir(GOLD):
        if not name.endswith(".npz"):
            continue
        z = np.load(os.path.join(GOLD, name))
        scene, w, h = (int(v) for v in z["meta"][:3])
        assert abi.build_scene(scene, w, h) == z["blob"].tobytes(), name


def test_scene_blob_parse_assemble_roundtrip():
    for scene in (0, 1, 3):
        blob = abi.build_scene(scene, 40, 30)
        assert abi.assemble_scene(abi.parse_scene(blob)) == blob
    import oracle
    big = oracle.cluttered_cornell(40, 30, n_extra=12)
    assert abi.parse_scene(big)["header"].n_prims == abi.parse_scene(abi.build_scene(0, 40, 30))["header"].n_prims + 12
