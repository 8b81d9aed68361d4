"""Image exporters behind su_export_frame (reference core/exporting/image_sequence.zig + image/encoding/*): the encoded
files are decoded here with independent readers (zlib for PNG, struct for EXR, the Radiance RLE for .hdr)."""
import ctypes
import json
import os
import struct
import zlib

import numpy as np
import pytest

import zyg_amd
from zyg_amd import scenes


def _encode(lib, fmt, flags, img, crop=None):
    h, w, _ = img.shape
    crop = (ctypes.c_int32 * 4)(*(crop or (0, 0, w, h)))
    lib.su_ext_encode_image.restype = ctypes.c_uint64
    lib.su_ext_encode_image.argtypes = [ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p,
                                        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64]
    img = np.ascontiguousarray(img, dtype=np.float32)
    n = lib.su_ext_encode_image(fmt, flags, w, h, crop, img.ctypes.data, None, 0)
    buf = (ctypes.c_uint8 * n)()
    assert lib.su_ext_encode_image(fmt, flags, w, h, crop, img.ctypes.data, buf, n) == n
    return bytes(buf)


def _decode_png(data):
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    at, idat, ihdr = 8, b"", None
    while at < len(data):
        n, typ = struct.unpack(">I4s", data[at:at + 8])
        body = data[at + 8:at + 8 + n]
        assert zlib.crc32(typ + body) == struct.unpack(">I", data[at + 8 + n:at + 12 + n])[0]
        if typ == b"IHDR":
            ihdr = struct.unpack(">IIBBBBB", body)
        elif typ == b"IDAT":
            idat += body
        at += 12 + n
    w, h, depth, ctype = ihdr[:4]
    nc = {2: 3, 6: 4, 0: 1}[ctype]
    raw = np.frombuffer(zlib.decompress(idat), dtype=np.uint8).reshape(h, w * nc + 1)
    assert depth == 8 and np.all(raw[:, 0] == 0)
    return raw[:, 1:].reshape(h, w, nc)


def _gamma(x):
    x = np.asarray(x, dtype=np.float32)
    return np.where(x <= 0, 0, np.where(x < 0.0031308, 12.92 * x, np.where(x < 1, 1.055 * np.power(x, np.float32(1 / 2.4)) - 0.055, 1)))


def _test_image(h=12, w=20):
    rng = np.random.default_rng(5)
    img = rng.uniform(0, 1.3, (h, w, 4)).astype(np.float32)
    img[0, 0] = (0, 0, 0, 0)
    img[1, 1] = (5.0, 0.001, 1.0, 1.0)
    return img


@pytest.mark.parametrize("alpha", [False, True])
def test_png_srgb8(product_lib, alpha):
    img = _test_image()
    px = _decode_png(_encode(product_lib, 0, 1 if alpha else 0, img))
    want = (_gamma(img[..., :3]) * 255.0 + 0.5).astype(np.uint8)
    assert np.abs(px[..., :3].astype(int) - want.astype(int)).max() <= 1  # pow of libm vs numpy at the .5 boundary
    assert np.mean(px[..., :3] == want) > 0.99
    if alpha:
        assert np.array_equal(px[..., 3], (np.minimum(img[..., 3], 1.0) * 255.0 + 0.5).astype(np.uint8))
    # the crop window: zero outside (srgb.zig:86-88)
    px = _decode_png(_encode(product_lib, 0, 0, img, crop=(2, 1, 10, 7)))
    assert px[0].max() == 0 and px[:, :2].max() == 0 and px[7:].max() == 0 and px[:, 10:].max() == 0 and px[1:7, 2:10].max() > 0


def test_png_error_diffusion(product_lib):
    """Error diffusion (srgb.zig:130-160): a flat value between two codes dithers to the right mean along each row."""
    img = np.zeros((8, 256, 4), dtype=np.float32)
    lin = 0.2
    img[..., :3] = lin
    px = _decode_png(_encode(product_lib, 0, 2, img))
    exact = float(_gamma(lin)) * 255.0
    assert set(np.unique(px)) <= {int(np.floor(exact)), int(np.ceil(exact))}
    assert abs(px[..., 0].astype(np.float64).mean(axis=1) - exact).max() < 0.02
    plain = _decode_png(_encode(product_lib, 0, 0, img))
    assert len(np.unique(plain)) == 1


@pytest.mark.parametrize("half,alpha", [(True, False), (False, True)])
def test_exr_scanlines(product_lib, half, alpha):
    img = _test_image()
    h, w, _ = img.shape
    crop = (1, 2, w - 3, h - 1)
    data = _encode(product_lib, 1, (1 if alpha else 0) | (0 if half else 4), img, crop=crop)
    assert data[:4] == bytes([0x76, 0x2f, 0x31, 0x01]) and data[4:8] == bytes([2, 0, 0, 0])
    at, attrs = 8, {}
    while data[at] != 0:
        name_end = data.index(b"\0", at)
        type_end = data.index(b"\0", name_end + 1)
        size = struct.unpack("<I", data[type_end + 1:type_end + 5])[0]
        attrs[data[at:name_end].decode()] = data[type_end + 5:type_end + 5 + size]
        at = type_end + 5 + size
    at += 1
    assert attrs["compression"] == b"\0" and struct.unpack("<4i", attrs["dataWindow"]) == (crop[0], crop[1], crop[2] - 1, crop[3] - 1)
    assert struct.unpack("<4i", attrs["displayWindow"]) == (0, 0, w - 1, h - 1)
    names = [c for c in ("A", "B", "G", "R") if alpha or c != "A"]
    ch, at2 = [], 0
    while attrs["channels"][at2] != 0:
        e = attrs["channels"].index(b"\0", at2)
        ch.append((attrs["channels"][at2:e].decode(), struct.unpack("<I", attrs["channels"][e + 1:e + 5])[0]))
        at2 = e + 17
    assert ch == [(n, 1 if half else 2) for n in names]
    cw, chh = crop[2] - crop[0], crop[3] - crop[1]
    offsets = struct.unpack("<%dQ" % chh, data[at:at + 8 * chh])
    dt = np.float16 if half else np.float32
    for row, off in enumerate(offsets):
        y, nbytes = struct.unpack("<iI", data[off:off + 8])
        assert y == crop[1] + row and nbytes == cw * len(names) * dt().itemsize
        vals = np.frombuffer(data[off + 8:off + 8 + nbytes], dtype=dt).reshape(len(names), cw)
        for k, n in enumerate(names):
            want = img[y, crop[0]:crop[2], "RGBA".index(n)].astype(dt)  # numpy rounds to nearest even, like @floatCast
            assert np.array_equal(vals[k], want), (row, n)


def _decode_rgbe(data):
    end = data.index(b"\n\n") + 2
    assert data.startswith(b"#?RGBE\nFORMAT=32-bit_rle_rgbe")
    line_end = data.index(b"\n", end)
    _, h, _, w = data[end:line_end].split()
    h, w = int(h), int(w)
    at = line_end + 1
    out = np.zeros((h, w, 4), dtype=np.uint8)
    for y in range(h):
        assert data[at] == 2 and data[at + 1] == 2 and (data[at + 2] << 8 | data[at + 3]) == w
        at += 4
        for k in range(4):
            x = 0
            while x < w:
                n = data[at]
                if n > 128:
                    out[y, x:x + n - 128, k] = data[at + 1]
                    x += n - 128
                    at += 2
                else:
                    out[y, x:x + n, k] = np.frombuffer(data[at + 1:at + 1 + n], dtype=np.uint8)
                    x += n
                    at += 1 + n
            assert x == w
    assert at == len(data)
    return out


def test_rgbe_rle(product_lib):
    img = _test_image(6, 40)
    img[2, 5:30, :3] = (0.25, 0.5, 0.75)  # a long run in every plane
    img[3] = 0.0
    rgbe = _decode_rgbe(_encode(product_lib, 2, 0, img))
    m, e = np.frexp(img[..., :3].max(axis=-1))
    scale = np.where(img[..., :3].max(axis=-1) < 1e-32, 0, m * 256.0 / np.maximum(img[..., :3].max(axis=-1), 1e-38))
    want = (img[..., :3] * scale[..., None]).astype(np.uint8)
    assert np.array_equal(rgbe[..., :3], want)
    assert np.array_equal(rgbe[..., 3], np.where(scale > 0, e + 128, 0).astype(np.uint8))
    decoded = rgbe[..., :3].astype(np.float64) * np.exp2(rgbe[..., 3:4].astype(np.float64) - 136.0)
    np.testing.assert_allclose(decoded[img[..., :3].max(axis=-1) > 0.05], img[..., :3][img[..., :3].max(axis=-1) > 0.05], atol=0.012, rtol=0.02)


@pytest.mark.gpu
def test_su_export_frame_writes_the_sequence(product_lib, tmp_path):
    """su_exporters_create + su_export_frame (capi.zig:141-160, 556-566): image_00_000003.png / .exr / .hdr of the resolved
    frame, named like the reference's image sequence (image_sequence.zig:37-42)."""
    w, h, spp = 48, 32, 4
    os.environ["ZYG_EXPORT_DIR"] = str(tmp_path)
    try:
        with zyg_amd.Engine(product_lib) as e:
            scenes.cornell(e, w, h, spp)
            e.sensor(json.dumps({"tonemapper": {"ACES": {"exposure": 0.0}}}))
            lib = e.lib
            for cfg in ({"Image": {"format": "PNG", "error_diffusion": True}}, {"Image": {"format": "EXR", "bitdepth": 32}},
                        {"Image": {"format": "RGBE"}}):
                assert lib.su_exporters_create(json.dumps(cfg).encode()) == 0
                e.start_frame(3)
                e.render_iterations(spp)
                assert lib.su_export_frame() == 0
            img = e.resolve(w, h)
        png = _decode_png(open(tmp_path / "image_00_000003.png", "rb").read())
        assert png.shape == (h, w, 3) and abs(png.astype(np.float64) - _gamma(img[..., :3]) * 255.0).max() <= 1.0
        assert (tmp_path / "image_00_000003.exr").stat().st_size > w * h * 12
        hdr = _decode_rgbe(open(tmp_path / "image_00_000003.hdr", "rb").read())
        assert hdr.shape == (h, w, 4)
    finally:
        del os.environ["ZYG_EXPORT_DIR"]
