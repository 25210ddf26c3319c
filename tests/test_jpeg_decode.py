"""The host's own baseline JPEG decoder (raytracing_weekend_amd/host/JpegDecode.h, SURVEY 8f rank 1 "stb image load") against
Pillow's decoder (libjpeg-turbo) on files written here: every chroma layout, grey, odd sizes, restart intervals, quality
levels. Decoders may differ by rounding in the inverse DCT and the colour conversion; chroma upsampling adds a little where
chroma is subsampled. The reference's own asset is decoded too where the reference tree is present. CPU only."""
import ctypes as C
import io
import os

import numpy as np
import pytest

from raytracing_weekend_amd import abi

PIL = pytest.importorskip("PIL.Image")


def decode(data):
    lib = abi.load_host()
    lib.rtw_host_decode_jpeg.restype = C.c_int
    lib.rtw_host_decode_jpeg.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_void_p, C.c_size_t, C.c_char_p, C.c_size_t]
    w, h = C.c_int(0), C.c_int(0)
    err = C.create_string_buffer(200)
    rc = lib.rtw_host_decode_jpeg(data, len(data), C.byref(w), C.byref(h), None, 0, err, 200)
    if rc == -2:
        raise ValueError(err.value.decode())
    assert rc == -5  # size query
    out = np.empty((h.value, w.value, 3), np.uint8)
    assert lib.rtw_host_decode_jpeg(data, len(data), C.byref(w), C.byref(h), out.ctypes.data, out.size, err, 200) == 0
    return out


def picture(w, h, seed):
    """Smooth colour fields plus some edges and noise: exercises DC prediction, long AC runs and end-of-block early."""
    rs = np.random.RandomState(seed)
    y, x = np.mgrid[0:h, 0:w].astype(np.float64)
    img = np.stack([128 + 100 * np.sin(x / 17.0 + seed) * np.cos(y / 23.0), 128 + 90 * np.cos(x / 9.0) * np.sin(y / 31.0 + 1), 100 + 0.4 * x + 0.3 * y], axis=-1)
    img[h // 3:h // 2, w // 4:w // 2] = (250, 20, 30)
    img += rs.normal(0, 6, img.shape)
    return np.clip(img, 0, 255).astype(np.uint8)


@pytest.mark.parametrize("w,h", [(64, 48), (37, 29), (8, 8), (1, 1), (130, 17)])
@pytest.mark.parametrize("sub", [0, 1, 2])  # Pillow: 0 = 4:4:4, 1 = 4:2:2, 2 = 4:2:0
def test_colour_layouts_match_pillow(w, h, sub):
    src = picture(w, h, sub)
    for quality in (95, 60):
        buf = io.BytesIO()
        PIL.fromarray(src).save(buf, format="JPEG", quality=quality, subsampling=sub)
        data = buf.getvalue()
        ref = np.asarray(PIL.open(io.BytesIO(data)).convert("RGB"))
        got = decode(data)
        assert got.shape == ref.shape
        d = np.abs(got.astype(np.int32) - ref.astype(np.int32))
        # same file, two decoders: rounding of the IDCT / colour matrix (<= 2), and where chroma is subsampled the upsampling
        # filters may differ on sharp colour edges
        assert np.percentile(d, 99) <= (2 if sub == 0 else 6) and d.mean() < (0.6 if sub == 0 else 1.5), (d.max(), d.mean())
        assert d.max() <= (3 if sub == 0 else 48)


def test_grey_and_restart_intervals():
    src = picture(100, 70, 5)
    buf = io.BytesIO()
    PIL.fromarray(src).convert("L").save(buf, format="JPEG", quality=90)
    ref = np.asarray(PIL.open(io.BytesIO(buf.getvalue())).convert("RGB"))
    got = decode(buf.getvalue())
    assert np.abs(got.astype(int) - ref.astype(int)).max() <= 2 and np.array_equal(got[..., 0], got[..., 1])
    for sub in (0, 2):
        buf = io.BytesIO()
        PIL.fromarray(src).save(buf, format="JPEG", quality=85, subsampling=sub, restart_marker_blocks=3)
        data = buf.getvalue()
        assert b"\xff\xdd" in data  # a DRI segment
        ref = np.asarray(PIL.open(io.BytesIO(data)).convert("RGB"))
        got = decode(data)
        d = np.abs(got.astype(int) - ref.astype(int))
        assert np.percentile(d, 99) <= (2 if sub == 0 else 6)


def test_unsupported_and_corrupt_files_are_refused():
    src = picture(40, 40, 1)
    buf = io.BytesIO()
    PIL.fromarray(src).save(buf, format="JPEG", quality=90, progressive=True)
    with pytest.raises(ValueError, match="progressive"):
        decode(buf.getvalue())
    with pytest.raises(ValueError):
        decode(b"P6\n1 1\n255\n\x00\x00\x00")
    buf = io.BytesIO()
    PIL.fromarray(src).save(buf, format="JPEG", quality=90)
    good = buf.getvalue()
    rs = np.random.RandomState(0)
    for _ in range(200):  # truncations and byte flips: any answer but a crash
        b = bytearray(good[:rs.randint(4, len(good))]) if rs.rand() < 0.5 else bytearray(good)
        for _ in range(rs.randint(1, 4)):
            b[rs.randint(2, len(b))] = rs.randint(0, 256)
        try:
            decode(bytes(b))
        except ValueError:
            pass


def test_truncated_scan_and_header_bombs_are_refused():
    """a scan that ends early must fail (not decode zeros to the end), and a header that declares more blocks than
    the file could hold must be refused before any plane is allocated."""
    import struct
    src = picture(96, 80, 3)
    buf = io.BytesIO()
    PIL.fromarray(src).save(buf, format="JPEG", quality=90, subsampling=0)
    good = buf.getvalue()
    sos = good.index(b"\xff\xda")
    for cut in (sos + 20, sos + (len(good) - sos) // 2, len(good) - 40):
        with pytest.raises(ValueError, match="truncated"):
            decode(good[:cut])
    # the entropy-coded data cut in the middle but with the EOI marker kept: a marker inside the data
    with pytest.raises(ValueError, match="truncated"):
        decode(good[:sos + (len(good) - sos) // 2] + b"\xff\xd9")
    # header bomb: the same file claiming 32768 x 32768 pixels (3 GB of planes before the fix)
    sof = good.index(b"\xff\xc0")
    bomb = bytearray(good)
    bomb[sof + 5:sof + 9] = struct.pack(">HH", 32768, 32768)
    with pytest.raises(ValueError, match="more blocks than the file"):
        decode(bytes(bomb))
    assert decode(good).shape == (80, 96, 3)


def test_reference_asset_decodes_like_pillow():
    path = "/root/reference/RestOfLife/assets/earthmap.jpg"
    if not os.path.exists(path):
        pytest.skip("reference tree not present (GPU box)")
    data = open(path, "rb").read()
    got = decode(data)
    ref = np.asarray(PIL.open(path).convert("RGB"))
    assert got.shape == ref.shape == (512, 1024, 3)
    d = np.abs(got.astype(int) - ref.astype(int))
    assert d.max() <= 3 and d.mean() < 0.5
    # and through the scene description: with RTW_ASSET_DIR pointing at the reference's assets, scene 2's image texture is that file
    os.environ["RTW_ASSET_DIR"] = os.path.dirname(path)
    try:
        parts = abi.parse_scene(abi.build_scene(2, 32, 32))
    finally:
        del os.environ["RTW_ASSET_DIR"]
    t = next(t for t in parts["textures"] if t.type == abi.TEX_IMAGE)
    words = np.frombuffer(parts["texdata"], np.uint32)
    assert (words[t.data], words[t.data + 1]) == (1024, 512)
    texels = words[t.data + 2:t.data + 2 + 1024 * 512].reshape(512, 1024)
    assert np.array_equal(texels[::-1] & 0xff, got[..., 0]) and np.array_equal((texels[::-1] >> 16) & 0xff, got[..., 2])  # rows flipped (ioTexture.h:247-250)
