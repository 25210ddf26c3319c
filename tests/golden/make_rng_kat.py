"""Generates tests/golden/rng_kat.json from the REFERENCE's own lib/random.cuh, compiled where it lies
by oracle/Makefile (target `ref`) into oracle/_ref/libref_random.so. Run in the build container only
(the reference tree does not travel); the JSON it writes is the committed fixture.

    make -C oracle ref && python tests/golden/make_rng_kat.py
"""
import ctypes as C
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lib = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_random.so"))
for f in (lib.ref_tea64, lib.ref_tea16, lib.ref_tea4):
    f.restype = C.c_uint32
    f.argtypes = [C.c_uint32, C.c_uint32]
lib.ref_xorshift32.restype = C.c_uint32
lib.ref_xorshift32.argtypes = [C.POINTER(C.c_uint32)]
lib.ref_randf.restype = C.c_float
lib.ref_randf.argtypes = [C.POINTER(C.c_uint32)]

pairs = [(0, 0), (1, 0), (1919, 0), (1920 * 1079 + 1919, 0), (7, 3), (0xFFFFFFFF, 0xFFFFFFFF), (123456789, 4095),
         (7680 * 4319 + 7679, 4095), (40000, 15), (0x6314759, 0x314759)]
kat = {"source": "RestOfLife/lib/random.cuh compiled with g++ (oracle/Makefile target ref)",
       "tea64": [[a, b, lib.ref_tea64(a, b)] for a, b in pairs],
       "tea16": [[a, b, lib.ref_tea16(a, b)] for a, b in pairs],
       "tea4": [[a, b, lib.ref_tea4(a, b)] for a, b in pairs],
       "xorshift32": {}, "randf": {}}
for seed in (0x6314759, 0x314759, 1, 0xFFFFFFFF, 0xDEADBEEF):
    s = C.c_uint32(seed)
    kat["xorshift32"][str(seed)] = [lib.ref_xorshift32(C.byref(s)) for _ in range(16)]
    s = C.c_uint32(seed)
    kat["randf"][str(seed)] = [float.hex(lib.ref_randf(C.byref(s))) for _ in range(16)]
json.dump(kat, open(os.path.join(ROOT, "tests", "golden", "rng_kat.json"), "w"), indent=1)
print("wrote rng_kat.json")
