import ctypes
import os

import numpy as np

import zyg_amd


def oracle_render(oracle, scene_desc, view_desc, iterations, threads=None, iteration0=0, y_range=None, stats=None,
                  samples_per_call=1, frame=0):
    """`frame`: the frame the descriptions were compiled for (it matters where a frame does not start on a tick of 1/60 s: a camera
    "shutter" with another frames_per_second)."""
    w, h = view_desc.resolution[0], view_desc.resolution[1]
    film = np.zeros((h, w, 4), dtype=np.float32)
    y0, y1 = y_range if y_range else (0, h)
    st = stats if stats is not None else zyg_amd.Stats()
    rc = oracle.zyg_oracle_render(ctypes.addressof(scene_desc), ctypes.addressof(view_desc), frame, iteration0, iterations,
                                  samples_per_call, y0, y1, film.ctypes.data, threads or (os.cpu_count() or 1),
                                  ctypes.byref(st))
    assert rc == 0
    return film


def oracle_render_alpha(oracle, scene_desc, view_desc, iterations, threads=None):
    """Transparent sensor: returns (film, alpha plane)."""
    w, h = view_desc.resolution[0], view_desc.resolution[1]
    film = np.zeros((h, w, 4), dtype=np.float32)
    alpha = np.zeros((h, w), dtype=np.float32)
    st = zyg_amd.Stats()
    oracle.zyg_oracle_render_alpha.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32,
                                               ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p,
                                               ctypes.c_void_p, ctypes.c_int32, ctypes.POINTER(zyg_amd.Stats)]
    rc = oracle.zyg_oracle_render_alpha(ctypes.addressof(scene_desc), ctypes.addressof(view_desc), 0, 0, iterations, 1, 0, h,
                                        film.ctypes.data, alpha.ctypes.data, threads or (os.cpu_count() or 1), ctypes.byref(st))
    assert rc == 0
    return film, alpha


AOV_NAMES = ["Albedo", "Depth", "MaterialId", "GeometricNormal", "ShadingNormal", "Roughness", "Emission", "Direct", "Indirect"]


def oracle_render_aovs(oracle, scene_desc, view_desc, iterations, threads=None):
    """Returns (film, {class: raw plane}) for the classes view_desc.aov_slots names; planes cleared as aov_buffer.zig:36-46."""
    w, h = view_desc.resolution[0], view_desc.resolution[1]
    film = np.zeros((h, w, 4), dtype=np.float32)
    planes = {}
    ptrs = (ctypes.c_void_p * 9)()
    for k in range(9):
        if view_desc.aov_slots & (1 << k):
            planes[k] = np.zeros((h, w, 4), dtype=np.float32)
            if k == 1:
                planes[k][..., :3] = np.finfo(np.float32).max
            ptrs[k] = planes[k].ctypes.data
    st = zyg_amd.Stats()
    oracle.zyg_oracle_render_aovs.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32,
                                              ctypes.c_uint32, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p,
                                              ctypes.c_void_p, ctypes.c_int32, ctypes.POINTER(zyg_amd.Stats)]
    rc = oracle.zyg_oracle_render_aovs(ctypes.addressof(scene_desc), ctypes.addressof(view_desc), 0, 0, iterations, 1, 0, h,
                                       film.ctypes.data, None, ctypes.cast(ptrs, ctypes.c_void_p), threads or (os.cpu_count() or 1),
                                       ctypes.byref(st))
    assert rc == 0
    return film, planes


def oracle_resolve_aov(oracle, view_desc, aov_class, plane):
    h, w = plane.shape[:2]
    out = np.zeros_like(plane)
    oracle.zyg_oracle_resolve_aov.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32]
    assert oracle.zyg_oracle_resolve_aov(ctypes.addressof(view_desc), aov_class, plane.ctypes.data, out.ctypes.data, w * h) == 0
    return out


def oracle_resolve(oracle, view_desc, film):
    h, w = film.shape[:2]
    out = np.zeros_like(film)
    oracle.zyg_oracle_resolve(ctypes.addressof(view_desc), film.ctypes.data, out.ctypes.data, w * h)
    return out


def rms(a, b):
    """Per-pixel RMS over RGB of the resolved linear image, values clamped to [0, 1]."""
    a = np.clip(np.nan_to_num(a[..., :3]), 0.0, 1.0).astype(np.float64)
    b = np.clip(np.nan_to_num(b[..., :3]), 0.0, 1.0).astype(np.float64)
    return float(np.sqrt(np.mean((a - b) ** 2)))


def write_png(path, pixels, filters=(0,)):
    """A PNG of (H, W, C) uint8 pixels (C: 1 grey, 3 RGB, 4 RGBA), row y filtered with filters[y % len(filters)] (0 None, 1 Sub, 2 Up,
    3 Average, 4 Paeth), the image data in two IDAT chunks."""
    import struct
    import zlib
    h, w, c = pixels.shape
    raw = bytearray()
    prev = np.zeros(w * c, dtype=np.int32)
    for y in range(h):
        row = pixels[y].astype(np.int32).ravel()
        f = filters[y % len(filters)]
        left = np.concatenate([np.zeros(c, dtype=np.int32), row[:-c]])
        upleft = np.concatenate([np.zeros(c, dtype=np.int32), prev[:-c]])
        if 0 == f:
            pred = np.zeros_like(row)
        elif 1 == f:
            pred = left
        elif 2 == f:
            pred = prev
        elif 3 == f:
            pred = (left + prev) // 2
        else:
            p = left + prev - upleft
            pa, pb, pc = np.abs(p - left), np.abs(p - prev), np.abs(p - upleft)
            pred = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, prev, upleft))
        raw.append(f)
        raw.extend(((row - pred) & 255).astype(np.uint8).tobytes())
        prev = row
    data = zlib.compress(bytes(raw), 6)

    def chunk(kind, body):
        return struct.pack(">I", len(body)) + kind + body + struct.pack(">I", zlib.crc32(kind + body) & 0xFFFFFFFF)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, {1: 0, 3: 2, 4: 6}[c], 0, 0, 0)) +
                chunk(b"IDAT", data[:len(data) // 2]) + chunk(b"IDAT", data[len(data) // 2:]) + chunk(b"IEND", b""))
