#!/usr/bin/env python3
"""Writes a small SYNTHETIC dataset in the file layout of the Prague sky model's ground-level dataset
(thirdparty/arpraguesky/ArPragueSkyModelGround.c:119-291: arpragueskymodelground_read_radiance / _read_transmittance), so that the Sky
entity's bake (zyg_amd/csrc/host/sky_bake.hpp, sky.zig:172-269) runs end to end where the real SkyModelDataset.dat.gz (~100 MB,
not shipped with the reference) is absent.  The numbers are smooth functions that look like a sky (bluer towards the zenith and at
high visibility, an aureole around the sun, dimmer at low solar elevation); they are NOT the model's fitted coefficients.

    python3 tools/make_sky_dataset.py <directory>      ->  <directory>/sky/SkyModelDataset.dat.gz   (mount <directory> with su_mount)

Layout (little endian):
  radiance metadata: visibilities (int32) + values (f64), albedos, altitudes, elevations likewise; channels (int32), channel_start,
      channel_width (f64); tensor_components (int32); sun_nbreaks + breaks (f64), zenith_nbreaks + breaks, emph_nbreaks + breaks
  radiance data, for visibility / albedo / altitude / elevation / channel (innermost): per tensor component sun values (f16 x
      sun_nbreaks), zenith_scale (f64), zenith values (f16 x zenith_nbreaks); then emph values (f16 x emph_nbreaks)
  transmittance: n_d, n_a, visibilities, altitudes, rank (int32); altitudes (f32), visibilities (f32); U (f32: altitude x n_d x n_a x
      rank); V (f32: visibility x altitude x 11 wavelengths x rank)"""
import gzip
import os
import struct
import sys

import numpy as np


def dataset_bytes():
    out = bytearray()

    def ints(*v):
        out.extend(struct.pack("<%di" % len(v), *v))

    def f64(v):
        out.extend(np.asarray(v, dtype="<f8").tobytes())

    def f32(v):
        out.extend(np.asarray(v, dtype="<f4").tobytes())

    def f16(v):
        out.extend(np.asarray(v, dtype="<f2").tobytes())

    visibilities = [20.0, 60.0, 150.0]
    albedos = [0.0, 1.0]
    altitudes = [0.0]
    elevations = [-4.2, 0.0, 10.0, 30.0, 60.0, 90.0]  # degrees
    channels, channel_start, channel_width = 11, 320.0, 40.0
    tensor_components = 3
    sun_breaks = np.array([0.0, 0.05, 0.15, 0.4, 0.9, 1.6, 2.4, np.pi])
    zenith_breaks = np.array([0.0, 0.6, 1.1, 1.4, np.pi / 2, 2.0, np.pi])
    emph_breaks = np.array([0.0, 1.2, np.pi / 2, 2.2, np.pi])
    for values in (visibilities, albedos, altitudes, elevations):
        ints(len(values))
        f64(values)
    ints(channels)
    f64([channel_start, channel_width])
    ints(tensor_components)
    for breaks in (sun_breaks, zenith_breaks, emph_breaks):
        ints(len(breaks))
        f64(breaks)
    for vis in visibilities:
        haze = 20.0 / vis  # 1 at the haziest
        for albedo in albedos:
            for _altitude in altitudes:
                for elevation in elevations:
                    day = max(0.02, np.sin(np.radians(max(elevation, 0.0))) * 0.9 + 0.1)
                    for ch in range(channels):
                        wl = channel_start + (ch + 0.5) * channel_width
                        rayleigh = (550.0 / wl) ** 4
                        level = 0.02 * day * (0.35 + 0.65 * rayleigh * (1.0 - 0.5 * haze)) * (1.0 + 0.15 * albedo)
                        # sum over components of sun_t(gamma) * zenith_t(alpha), times emph(theta)
                        comps = (
                            (np.full_like(sun_breaks, level), 1.0, 1.0 + 0.8 * (zenith_breaks / (np.pi / 2)) ** 2),          # base, brighter at the horizon
                            (level * (2.0 + 6.0 * haze) * np.exp(-sun_breaks * 3.0), 2.0, 2.0 * np.ones_like(zenith_breaks)),  # the aureole (zenith_scale 2 undoes the 2)
                            (level * 0.5 * np.cos(sun_breaks) ** 2, 1.0, np.cos(np.minimum(zenith_breaks, np.pi / 2)) + 0.2),   # Rayleigh phase lobe
                        )
                        for sun_values, zenith_scale, zenith_values in comps:
                            f16(sun_values)
                            f64([zenith_scale])
                            f16(zenith_values)
                        f16(np.where(emph_breaks <= np.pi / 2, 1.0, 0.05))  # nothing below the horizon
    # transmittance: (sum_i U[a, d, i] * V[vis, wl, i])^2, a = cbrt(height / 100 km), d = (distance / max)^(1/4)
    n_d, n_a, rank = 8, 4, 2
    ints(n_d, n_a, len(visibilities), 1, rank)
    f32([0.0])
    f32(visibilities)
    u = np.zeros((1, n_d, n_a, rank))
    for d in range(n_d):
        for a in range(n_a):
            path = ((d + 0.5) / n_d) ** 4 * (1.0 - 0.6 * (a + 0.5) / n_a)  # optical path, shorter from higher up
            u[0, d, a] = (1.0, path)
    f32(u)
    v = np.zeros((len(visibilities), 1, 11, rank))
    for i, vis in enumerate(visibilities):
        for wl_i in range(11):
            wl = 340.0 + 40.0 * wl_i
            tau = (0.9 * (550.0 / wl) ** 4 + 0.6 * 20.0 / vis)  # Rayleigh + haze at the longest path
            v[i, 0, wl_i] = (1.0, -(1.0 - np.exp(-0.5 * tau)))
    f32(v)
    return bytes(out)


def write(directory):
    path = os.path.join(directory, "sky")
    os.makedirs(path, exist_ok=True)
    path = os.path.join(path, "SkyModelDataset.dat.gz")
    with gzip.open(path, "wb", compresslevel=6) as f:
        f.write(dataset_bytes())
    return path


if __name__ == "__main__":
    print(write(sys.argv[1] if len(sys.argv) > 1 else "."))
