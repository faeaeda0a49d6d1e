"""Deterministic synthetic PBR maps for the benchmark / parity workloads (seed 0x5EED).

There is no network for real texture sets, and the reference ships 2048x2048 PNGs that cannot travel
to the GPU box; these maps have the same shape (2048x2048 RGBA8 x 5 used maps) and exercise the same
code paths: value-noise albedo, roughness in [0.15, 0.95] (never 0 -- brdf.glsl's 0/0 hazard),
metallic in {0,1} blocks, ao in [0.5, 1], normals = small perturbations of (127,127,255).
Pure input data; no renderer arithmetic lives here.
"""
from __future__ import annotations

import numpy as np

SEED = 0x5EED


def _value_noise(rng, size, cells):
    """Bilinear value noise, tileable, in [0,1]; float64 -> deterministic across numpy versions that
    share the PCG64 stream."""
    g = rng.random((cells, cells))
    t = (np.arange(size) + 0.5) * (cells / size)
    i0 = np.floor(t).astype(np.int64) % cells
    i1 = (i0 + 1) % cells
    f = t - np.floor(t)
    f = f * f * (3 - 2 * f)
    a = g[i0][:, i0] * (1 - f)[None, :] + g[i0][:, i1] * f[None, :]
    b = g[i1][:, i0] * (1 - f)[None, :] + g[i1][:, i1] * f[None, :]
    return a * (1 - f)[:, None] + b * f[:, None]


def _u8(x):
    return np.clip(np.rint(x * 255.0), 0, 255).astype(np.uint8)


def _rgba(r, g, b):
    out = np.empty(r.shape + (4,), np.uint8)
    out[..., 0], out[..., 1], out[..., 2] = r, g, b
    out[..., 3] = 255
    return out


def make_material(size=2048, seed=SEED):
    """Returns dict name -> uint8 [size, size, 4] for albedo, metallic, roughness, ao, normal."""
    rng = np.random.Generator(np.random.PCG64(seed))
    base = _value_noise(rng, size, 16)
    fine = _value_noise(rng, size, 128)
    hue = _value_noise(rng, size, 8)
    alb_r = 0.25 + 0.7 * (0.6 * base + 0.4 * fine)
    alb_g = 0.20 + 0.7 * (0.5 * base + 0.5 * hue)
    alb_b = 0.15 + 0.7 * (0.7 * hue + 0.3 * fine)
    albedo = _rgba(_u8(alb_r), _u8(alb_g), _u8(alb_b))
    rough = 0.15 + 0.80 * _value_noise(rng, size, 32)
    r8 = np.maximum(_u8(rough), 39)  # 39/255 = 0.153 > 0.15
    roughness = _rgba(r8, r8, r8)
    blocks = max(size // 8, 1)
    cells = (rng.random((8, 8)) > 0.5).astype(np.float64)
    m = np.kron(cells, np.ones((blocks, blocks)))[:size, :size]
    m8 = _u8(m)
    metallic = _rgba(m8, m8, m8)
    a8 = _u8(0.5 + 0.5 * _value_noise(rng, size, 64))
    ao = _rgba(a8, a8, a8)
    nx = (_value_noise(rng, size, 96) - 0.5) * 0.35
    ny = (_value_noise(rng, size, 96) - 0.5) * 0.35
    nz = np.sqrt(np.clip(1.0 - nx * nx - ny * ny, 0.0, 1.0))
    normal = _rgba(_u8(nx * 0.5 + 0.5), _u8(ny * 0.5 + 0.5), _u8(nz * 0.5 + 0.5))
    return {"albedo": albedo, "metallic": metallic, "roughness": roughness, "ao": ao, "normal": normal}
