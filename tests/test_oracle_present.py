"""Presentation step of the oracle: binary16 HDR attachment, hdr_tone_mapping.frag:9-18, sRGB
encode + UNORM8.  The reference holds no vectors for it (it is driver work), so the pins are: numpy's own binary16
conversion, exp in binary64, the sRGB definition evaluated in binary64, and the frozen fixture presented.npz."""
import hashlib
import json
import os

import numpy as np

from conftest import GOLDEN
from oracle import bbo


def test_half_round_is_numpy_float16_round_trip():
    rng = np.random.default_rng(7)
    xs = np.concatenate([rng.standard_normal(4000).astype(np.float32) * s for s in (1e-8, 1e-6, 1e-4, 1, 100, 30000, 70000)] +
                        [np.array([0, -0.0, 65504, 65519.99, 65520, 65536, 1e-8, 5.96e-8, 2.98e-8, 2.9802322e-8, 6.1e-5,
                                   np.inf, -np.inf], np.float32)])
    with np.errstate(over="ignore"):
        want = xs.astype(np.float16).astype(np.float32)
    assert np.array_equal(bbo.half_round(xs).view(np.uint32), want.view(np.uint32))
    assert np.isnan(bbo.half_round(np.float32("nan")))


def test_exp_fixed_sequence_is_within_one_ulp():
    rng = np.random.default_rng(8)
    xs = np.concatenate([rng.uniform(-87, 0, 20000), rng.uniform(0, 88.7, 5000), rng.uniform(-1, 1, 5000)]).astype(np.float32)
    got = bbo.exp(xs).astype(np.float64)
    ref = np.exp(xs.astype(np.float64))
    ulp = np.spacing(ref.astype(np.float32)).astype(np.float64)
    assert (np.abs(got - ref) / ulp).max() < 1.0
    assert bbo.exp(np.float32(0.0)) == 1.0 and bbo.exp(np.float32(-200.0)) == 0.0
    assert np.isinf(bbo.exp(np.float32(100.0))) and np.isnan(bbo.exp(np.float32("nan")))


def _srgb_encode(c):  # IEC 61966-2-1, binary64
    c = np.asarray(c, np.float64)
    return np.where(c <= 0.0031308, 12.92 * c, 1.055 * np.power(np.maximum(c, 0), 1 / 2.4) - 0.055)


def test_srgb_byte_is_the_rounded_ideal_curve():
    thr = bbo.srgb_thresholds().astype(np.float64)
    assert np.all(np.diff(thr) > 0)
    # each threshold sits where round(255 * encode(c)) steps from k-1 to k
    k = np.arange(1, 256)
    assert np.allclose(255 * _srgb_encode(thr), k - 0.5, atol=2e-4)
    rng = np.random.default_rng(9)
    c = np.concatenate([rng.uniform(0, 1, 50000), rng.uniform(0, 0.01, 20000)]).astype(np.float32)
    rgba = np.zeros((c.size, 4), np.float32)
    rgba[:, 0] = c
    got = bbo.present(rgba, 0, 1.0, hdr16=False)[:, 0].astype(np.int64)
    e = 255 * _srgb_encode(c)
    want = np.floor(e + 0.5).astype(np.int64)
    away = np.abs(e + 0.5 - np.round(e + 0.5)) > 1e-3  # not within float noise of a rounding boundary
    assert np.array_equal(got[away], want[away])
    assert np.abs(got - want).max() <= 1


def test_present_edge_values_and_alpha():
    px = np.array([[0, 1, 2, 0], [-1, np.nan, np.inf, 0.5], [1e-9, 0.5, 0.9999, 7]], np.float32)
    out = bbo.present(px, 0, 1.0)
    assert out[0].tolist() == [0, 255, 255, 255]
    assert out[1].tolist() == [0, 0, 255, 255]       # negative and NaN -> 0, inf -> 255; alpha always 255
    assert out[2, 0] == 0 and out[2, 1] == 188 and out[2, 2] == 255   # linear 0.5 -> sRGB 188


def test_tone_map_on_off_and_fp16_stage():
    rng = np.random.default_rng(10)
    hdr = rng.uniform(0, 4, (1000, 4)).astype(np.float32)
    off = bbo.present(hdr, 0, 1.3)
    assert np.array_equal(off, bbo.present(bbo.half_round(hdr), 0, 1.3, hdr16=False))
    on = bbo.present(hdr, 1, 1.3)
    mapped = bbo.tone_map(bbo.half_round(hdr), 1, 1.3)      # the float-only entry point uses the same exp
    assert np.array_equal(on, bbo.present(mapped, 0, 1.0, hdr16=False))
    assert np.all(on[:, :3] <= 255) and np.all(on[:, 3] == 255)


def test_frozen_fixture():
    info = json.load(open(os.path.join(GOLDEN, "presented.json")))
    z = np.load(os.path.join(GOLDEN, "presented.npz"))
    thr = bbo.srgb_thresholds()
    assert np.array_equal(thr.view(np.uint32), z["srgb_thresholds_bits"])
    assert hashlib.sha256(thr.tobytes()).hexdigest() == info["thresholds_sha256"]
    hdr = np.load(os.path.join(GOLDEN, "oracle_frames.npz"))["c2_160x90_rgba_bits"].view(np.float32)
    for tag in ("plain", "tonemapped", "tonemapped_fp32"):
        i = info[tag]
        img = bbo.present(hdr, i["enable"], i["exposure"], i["hdr16"])
        assert np.array_equal(img, z[f"c2_160x90_{tag}"])
        assert hashlib.sha256(img.tobytes()).hexdigest() == i["sha256"]
