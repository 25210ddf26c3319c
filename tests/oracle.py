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
