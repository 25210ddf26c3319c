"""ctypes mirror of include/rtw.h and loaders for the in-tree shared libraries.

Plumbing only: scene blobs come from the C++ host description (librtw_host.so), rendering is
done by the HIP library (librtw_hip.so) through its C ABI. There is no Python or CPU fallback:
`load_hip()` raises if the HIP library has not been built.
"""
import ctypes as C
import os

import numpy as np

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
REPO_DIR = os.path.dirname(PKG_DIR)
HOST_LIB = os.path.join(PKG_DIR, "host", "librtw_host.so")
HIP_LIB = os.environ.get("RTW_HIP_LIB") or os.path.join(PKG_DIR, "csrc", "librtw_hip.so")

RTW_ABI_VERSION = 2
RTW_SCENE_VERSION = 1
RTW_SCENE_MAGIC = 0x57545221
RTW_RNG_PHILOX = 0
RTW_RNG_TEA_LCG = 1
RTW_EST_REFERENCE, RTW_EST_CORRECTED, RTW_EST_CORRECTED_NO_NEE, RTW_EST_MIXTURE = range(4)

# rtw_prim_type
PRIM_SPHERE, PRIM_MOVING_SPHERE, PRIM_RECT_X, PRIM_RECT_Y, PRIM_RECT_Z, PRIM_VOLUME_BOX, PRIM_VOLUME_SPHERE = range(7)
# rtw_material_type
MAT_LAMBERTIAN, MAT_DIFFUSE_LIGHT, MAT_METAL, MAT_DIELECTRIC, MAT_ISOTROPIC, MAT_NORMAL = range(6)
# rtw_pdf_gen
RTW_PDF_COSINE, RTW_PDF_MIXTURE_BIAS, RTW_PDF_MIXTURE, RTW_PDF_RECT_X, RTW_PDF_RECT_Y, RTW_PDF_RECT_Z = range(6)
# rtw_camera_type
RTW_CAM_PERSPECTIVE, RTW_CAM_ENVIRONMENT, RTW_CAM_ORTHOGRAPHIC = range(3)
# rtw_texture_type
TEX_CHECKER, TEX_CONSTANT, TEX_IMAGE, TEX_NOISE, TEX_NULL = range(5)


class Prim(C.Structure):
    _fields_ = [("type", C.c_int32), ("material", C.c_int32), ("xform", C.c_int32), ("flip", C.c_int32),
                ("p", C.c_float * 12)]


class Xform(C.Structure):
    _fields_ = [("m", C.c_float * 12), ("inv", C.c_float * 12)]


class Material(C.Structure):
    _fields_ = [("type", C.c_int32), ("texture", C.c_int32), ("fuzz_or_eta", C.c_float), ("bsdf_eval", C.c_int32)]


class Texture(C.Structure):
    _fields_ = [("type", C.c_int32), ("color", C.c_float * 3), ("odd", C.c_int32), ("even", C.c_int32),
                ("scale", C.c_float), ("data", C.c_uint32)]


class Light(C.Structure):
    _fields_ = [("position", C.c_float * 3), ("vec_u", C.c_float * 3), ("vec_v", C.c_float * 3),
                ("normal", C.c_float * 3), ("area", C.c_float), ("emission", C.c_float * 3)]


class Pdf(C.Structure):
    _fields_ = [("gen", C.c_int32), ("p0_gen", C.c_int32), ("p1_gen", C.c_int32), ("flip", C.c_int32),
                ("rect", C.c_float * 5), ("bias", C.c_float)]


class Camera(C.Structure):
    _fields_ = [("origin", C.c_float * 3), ("u", C.c_float * 3), ("v", C.c_float * 3), ("w", C.c_float * 3),
                ("lower_left", C.c_float * 3), ("horizontal", C.c_float * 3), ("vertical", C.c_float * 3),
                ("lens_radius", C.c_float), ("time0", C.c_float), ("time1", C.c_float)]


class SceneHeader(C.Structure):
    _fields_ = [("magic", C.c_uint32), ("version", C.c_uint32), ("total_bytes", C.c_uint32),
                ("n_prims", C.c_uint32), ("n_xforms", C.c_uint32), ("n_materials", C.c_uint32),
                ("n_textures", C.c_uint32), ("n_lights", C.c_uint32),
                ("off_prims", C.c_uint32), ("off_xforms", C.c_uint32), ("off_materials", C.c_uint32),
                ("off_textures", C.c_uint32), ("off_lights", C.c_uint32),
                ("sky_light", C.c_int32), ("off_texdata", C.c_uint32), ("texdata_bytes", C.c_uint32),
                ("camera", Camera), ("pdf", Pdf), ("camera_type", C.c_int32), ("reserved", C.c_uint32)]


class Params(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("spp", C.c_int32), ("max_depth", C.c_int32),
                ("seed", C.c_uint32), ("row0", C.c_int32), ("row1", C.c_int32), ("rng_kind", C.c_int32),
                ("sample_offset", C.c_int32), ("samples_per_pass", C.c_int32), ("row_stride", C.c_int32), ("estimator", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [("samples", C.c_uint64), ("segments", C.c_uint64), ("shadow_rays", C.c_uint64),
                ("algorithmic_bytes", C.c_uint64), ("bounce_launches", C.c_uint64), ("reserved", C.c_uint64),
                ("seconds", C.c_double), ("bounce_seconds", C.c_double),
                ("kernel_seconds", C.c_double * 5), ("kernel_launches", C.c_uint64 * 5), ("kernel_segments", C.c_uint64 * 5)]

    KERNELS = ("k_first", "k_shade", "k_trace", "k_bounce", "k_path")

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_ if k != "reserved"}


HIP_SYMBOLS = ["rtw_abi_version", "rtw_create", "rtw_upload_scene", "rtw_render", "rtw_render_device",
               "rtw_destroy", "rtw_last_error", "rtw_debug_intersect", "rtw_denoise"]

_host = None
_hip = None


def load_host():
    global _host
    if _host is None:
        if not os.path.exists(HOST_LIB):
            raise RuntimeError(f"{HOST_LIB} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
        lib = C.CDLL(HOST_LIB)
        lib.rtw_host_build_scene.restype = C.c_int
        lib.rtw_host_build_scene.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
        _host = lib
    return _host


def load_hip():
    """Load librtw_hip.so. Fails loudly when it is absent: there is no fallback path."""
    global _hip
    if _hip is None:
        if not os.path.exists(HIP_LIB):
            raise RuntimeError(f"{HIP_LIB} is missing: the HIP extension must be built "
                               "(`python -c 'import __graft_entry__ as g; g.build()'`); no CPU fallback exists")
        lib = C.CDLL(HIP_LIB)
        lib.rtw_abi_version.restype = C.c_int
        lib.rtw_create.restype = C.c_int
        lib.rtw_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_int)]
        lib.rtw_upload_scene.restype = C.c_int
        lib.rtw_upload_scene.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        lib.rtw_render.restype = C.c_int
        lib.rtw_render.argtypes = [C.c_void_p, C.POINTER(Params), C.c_void_p, C.POINTER(Stats)]
        lib.rtw_render_device.restype = C.c_int
        lib.rtw_render_device.argtypes = [C.c_void_p, C.POINTER(Params), C.c_void_p, C.c_void_p, C.POINTER(Stats)]
        lib.rtw_destroy.restype = C.c_int
        lib.rtw_destroy.argtypes = [C.c_void_p]
        lib.rtw_last_error.restype = C.c_char_p
        lib.rtw_last_error.argtypes = [C.c_void_p]
        lib.rtw_denoise.restype = C.c_int
        lib.rtw_denoise.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_float]
        lib.rtw_debug_intersect.restype = C.c_int
        lib.rtw_debug_intersect.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        if lib.rtw_abi_version() != RTW_ABI_VERSION:
            raise RuntimeError("librtw_hip.so ABI version mismatch")
        _hip = lib
    return _hip


def build_scene(scene, nx, ny):
    """Scene blob (bytes) of reference scene 0/1/3 for an nx x ny image, from the C++ host description."""
    lib = load_host()
    need = C.c_size_t(0)
    rc = lib.rtw_host_build_scene(scene, nx, ny, None, 0, C.byref(need))
    if rc == -1:
        raise ValueError(f"unknown scene {scene}")
    buf = C.create_string_buffer(need.value)
    rc = lib.rtw_host_build_scene(scene, nx, ny, buf, need.value, C.byref(need))
    if rc != 0:
        raise RuntimeError(f"rtw_host_build_scene failed: {rc}")
    return buf.raw


def parse_scene(blob):
    """Decode a scene blob into ctypes views (header, prims, xforms, materials, textures, lights)."""
    h = SceneHeader.from_buffer_copy(blob[:C.sizeof(SceneHeader)])

    def arr(t, off, n):
        return (t * n).from_buffer_copy(blob[off:off + n * C.sizeof(t)])
    return {
        "header": h,
        "prims": arr(Prim, h.off_prims, h.n_prims),
        "xforms": arr(Xform, h.off_xforms, h.n_xforms),
        "materials": arr(Material, h.off_materials, h.n_materials),
        "textures": arr(Texture, h.off_textures, h.n_textures),
        "lights": arr(Light, h.off_lights, h.n_lights),
        "texdata": bytes(blob[h.off_texdata:h.off_texdata + h.texdata_bytes]) if h.off_texdata else b"",
    }


def assemble_scene(parts):
    """Inverse of parse_scene: serialise {"header", "prims", "xforms", "materials", "textures", "lights"[, "texdata"]}
    (ctypes arrays or lists of the structs, texdata = bytes of the texture data section) into a blob; counts, offsets
    and total_bytes are recomputed."""
    src = parts["header"]
    h = SceneHeader.from_buffer_copy(bytes(src))
    tables = [list(parts[k]) for k in ("prims", "xforms", "materials", "textures", "lights")]
    h.n_prims, h.n_xforms, h.n_materials, h.n_textures, h.n_lights = (len(t) for t in tables)
    sizes = [C.sizeof(t) for t in (Prim, Xform, Material, Texture, Light)]
    offs = [(C.sizeof(SceneHeader) + 15) // 16 * 16]
    for t, sz in zip(tables, sizes):
        offs.append((offs[-1] + len(t) * sz + 15) // 16 * 16)
    h.off_prims, h.off_xforms, h.off_materials, h.off_textures, h.off_lights = offs[:5]
    texdata = bytes(parts.get("texdata", b""))
    h.off_texdata, h.texdata_bytes = (offs[5], len(texdata)) if texdata else (0, 0)
    h.total_bytes = (offs[5] + len(texdata) + 15) // 16 * 16
    buf = bytearray(h.total_bytes)
    buf[offs[5]:offs[5] + len(texdata)] = texdata
    buf[0:C.sizeof(SceneHeader)] = bytes(h)
    for t, sz, off in zip(tables, sizes, offs):
        for i, obj in enumerate(t):
            buf[off + i * sz:off + (i + 1) * sz] = bytes(obj)
    return bytes(buf)


def make_params(width, height, spp, max_depth, seed=0x6314759, row0=0, row1=None, rng_kind=RTW_RNG_PHILOX,
                sample_offset=0, samples_per_pass=0, row_stride=0, estimator=0):
    p = Params()
    p.width, p.height, p.spp, p.max_depth = width, height, spp, max_depth
    p.seed = seed
    p.row0 = row0
    p.row1 = height if row1 is None else row1
    p.rng_kind = rng_kind
    p.sample_offset = sample_offset
    p.samples_per_pass = samples_per_pass
    p.row_stride = row_stride
    p.estimator = estimator
    return p


def local_rows(params):
    """Number of image rows a render with these params produces."""
    k = max(1, params.row_stride)
    return max(0, (params.row1 - params.row0 + k - 1) // k)


class Renderer:
    """Thin owner of one rtw_ctx: one GPU (device=i) or an in-library group (device=[i, j, ...]: rtw_create with
    n_devices > 1, one interleaved row shard per entry, gathered on the first device)."""

    def __init__(self, device=0):
        self.lib = load_hip()
        self.ctx = C.c_void_p()
        ids = list(device) if isinstance(device, (list, tuple)) else [device]
        dev = (C.c_int * len(ids))(*ids)
        rc = self.lib.rtw_create(C.byref(self.ctx), len(ids), dev)
        if rc != 0:
            raise RuntimeError(f"rtw_create failed: {rc}")

    def _check(self, rc, what):
        if rc != 0:
            msg = self.lib.rtw_last_error(self.ctx)
            raise RuntimeError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")

    def upload_scene(self, blob):
        self._check(self.lib.rtw_upload_scene(self.ctx, blob, len(blob)), "rtw_upload_scene")

    def render(self, params):
        rows = local_rows(params)
        out = np.empty((rows, params.width, 4), dtype=np.float32)
        st = Stats()
        self._check(self.lib.rtw_render(self.ctx, C.byref(params), out.ctypes.data, C.byref(st)), "rtw_render")
        return out, st

    def render_device(self, params, device_ptr, stream_ptr=0):
        st = Stats()
        self._check(self.lib.rtw_render_device(self.ctx, C.byref(params), C.c_void_p(device_ptr),
                                               C.c_void_p(stream_ptr), C.byref(st)), "rtw_render_device")
        return st

    def denoise(self, img, iterations=5, sigma=0.5):
        """rtw_denoise on an (h, w, 4) float32 image; returns the filtered image."""
        src = np.ascontiguousarray(img, dtype=np.float32)
        h, w = src.shape[:2]
        out = np.empty_like(src)
        self._check(self.lib.rtw_denoise(self.ctx, src.ctypes.data, out.ctypes.data, w, h, iterations, sigma), "rtw_denoise")
        return out

    def debug_intersect(self, rays, ray_time=None, gather_time=None):
        rays = np.ascontiguousarray(rays, dtype=np.float32)
        n = rays.shape[0]
        rt = None if ray_time is None else np.ascontiguousarray(ray_time, dtype=np.float32)
        gt = None if gather_time is None else np.ascontiguousarray(gather_time, dtype=np.float32)
        t = np.empty(n, dtype=np.float32)
        prim = np.empty(n, dtype=np.int32)
        self._check(self.lib.rtw_debug_intersect(self.ctx, rays.ctypes.data,
                                                 None if rt is None else rt.ctypes.data,
                                                 None if gt is None else gt.ctypes.data,
                                                 n, t.ctypes.data, prim.ctypes.data), "rtw_debug_intersect")
        return t, prim

    def close(self):
        if self.ctx:
            self.lib.rtw_destroy(self.ctx)
            self.ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
