"""raytracing_weekend_amd — MI355X-native wavefront path tracer behind the reference's Director surface.

csrc/  hand-written HIP kernels (gfx950) + the C ABI of include/rtw.h  -> librtw_hip.so
host/  C++ host surface mirroring the reference (InputParser, Director, scene/ description) -> rtw_render, librtw_host.so
abi.py ctypes plumbing used by bench.py and tests
"""
from . import abi  # noqa: F401
from .abi import Renderer, build_scene, make_params, parse_scene  # noqa: F401
