"""The C-ABI library loads on a CPU-only machine and exports every symbol include/rtw.h declares
(no compute calls here: those are the -m gpu tests)."""
import ctypes as C
import os
import re

from raytracing_weekend_amd import abi


def declared_functions():
    text = open(os.path.join(abi.REPO_DIR, "include", "rtw.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rtw_[a-z_]+)\s*\(", text)))


def test_header_declares_the_expected_entry_points():
    assert declared_functions() == sorted(abi.HIP_SYMBOLS)


def test_hip_library_exports_every_declared_symbol():
    assert os.path.exists(abi.HIP_LIB), "librtw_hip.so must be built in-tree (python __graft_entry__.py)"
    lib = C.CDLL(abi.HIP_LIB)
    for name in declared_functions():
        assert hasattr(lib, name), name
    lib.rtw_abi_version.restype = C.c_int
    assert lib.rtw_abi_version() == abi.RTW_ABI_VERSION


def test_host_library_loads():
    lib = abi.load_host()
    assert hasattr(lib, "rtw_host_build_scene")


def test_null_context_is_an_error_not_a_crash():
    lib = abi.load_hip()
    assert lib.rtw_upload_scene(None, None, 0) < 0
    assert lib.rtw_render(None, None, None, None) < 0
    assert lib.rtw_destroy(None) < 0
    assert lib.rtw_last_error(None) is not None


def test_no_exception_crosses_the_c_abi(monkeypatch):
    """include/rtw.h promises error codes, not exceptions. RTW_TEST_FAULT=entry:<kind> makes every entry point
    throw inside its guard before it looks at its arguments: a std::bad_alloc must come back as RTW_ERR_OOM, anything else
    as RTW_ERR_DEVICE, and the process must live (without the barrier std::terminate ends it). The in-thread and
    mid-upload injection points need a GPU: tests/test_gpu_round3.py."""
    lib = abi.load_hip()
    ctx = C.c_void_p()
    p = abi.make_params(8, 8, 1, 2)
    out = (C.c_float * 256)()
    for kind, want in (("bad_alloc", -5), ("runtime", -4)):
        monkeypatch.setenv("RTW_TEST_FAULT", "entry:" + kind)
        assert lib.rtw_create(C.byref(ctx), 1, None) == want and not ctx.value
        assert lib.rtw_upload_scene(None, b"x", 1) == want
        assert lib.rtw_render(None, C.byref(p), out, None) == want
        assert lib.rtw_render_device(None, C.byref(p), None, None, None) == want
        assert lib.rtw_denoise(None, None, None, 1, 1, 1, 1.0) == want
        assert lib.rtw_debug_intersect(None, None, None, None, 0, None, None) == want
        assert lib.rtw_destroy(None) == want
    monkeypatch.delenv("RTW_TEST_FAULT")
    assert lib.rtw_destroy(None) == -1  # back to the plain argument check
    # every extern "C" definition in the library's source runs inside the guard
    src = open(os.path.join(abi.PKG_DIR, "csrc", "rtw_hip.hip")).read()
    ext = src[src.index('extern "C" {'):]
    for name in declared_functions():
        body = ext[ext.index(name + "("):]
        body = body[:body.index("\n}") + 2] if "{" in body.split("\n")[0] else body
        if name in ("rtw_abi_version", "rtw_last_error"):
            continue  # no allocation, nothing to throw
        assert "guarded(" in body.split("\n}")[0], name


def test_product_package_does_not_reference_the_oracle():
    pkg = os.path.join(abi.REPO_DIR, "raytracing_weekend_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                text = open(os.path.join(root, f), errors="ignore").read()
                # comments may name the oracle; code must not load, call, include or import it
                assert "librtw_oracle" not in text and "rtwo_" not in text and "import oracle" not in text, f
                assert not re.search(r'#include\s*[<"][^>"]*oracle', text), f
