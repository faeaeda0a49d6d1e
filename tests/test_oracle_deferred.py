"""Deferred variant of the oracle: gbuffer.vert/.frag into binary16 attachments + brdf.frag on
every pixel.  The reference holds no vectors for it; pins are structural properties and the frozen fixture."""
import hashlib
import json
import os

import numpy as np

from conftest import GOLDEN
from bibim_renderer_amd import configs, textures
from oracle import bbo, scenes


def _scene(cfg=configs.C2, size=(160, 90), **kw):
    mat = bbo.MaterialData(textures.make_material(64))
    return scenes.shaderball_scene(cfg.scaled(size[0], size[1], 64), mat, **kw)


def test_frozen_fixture():
    z = np.load(os.path.join(GOLDEN, "deferred.npz"))
    info = json.load(open(os.path.join(GOLDEN, "deferred.json")))
    rgba, gbuf, prim, depth, st = bbo.render_deferred(_scene())
    lit = bbo.render_deferred(_scene(), flags=bbo.FLAG_LITERAL)[0]      # the statement-by-statement light loop, frozen too
    assert np.array_equal(lit.view(np.uint32), z["c2_160x90_literal_rgba_bits"])
    assert st == info["c2_160x90"]
    assert np.array_equal(rgba.view(np.uint32), z["c2_160x90_rgba_bits"])
    assert np.array_equal(gbuf, z["c2_160x90_gbuffer_f16"].astype(np.float32))
    assert np.array_equal(prim, z["c2_160x90_prim"]) and np.array_equal(depth.view(np.uint32), z["c2_160x90_depth_bits"])
    assert hashlib.sha256(rgba.tobytes()).hexdigest() == info["rgba_sha256"]


def test_gbuffer_is_binary16_and_laid_out_as_the_attachments():
    sc = _scene(configs.C3, (320, 180))
    rgba, g, prim, depth, st = bbo.render_deferred(sc)
    cov = prim != bbo.NO_PRIM
    with np.errstate(over="ignore"):
        assert np.array_equal(g, g.astype(np.float16).astype(np.float32))      # R16G16B16A16_SFLOAT, src/main.cpp:443
    assert np.all(g[~cov] == 0)                                                 # cleared, src/main.cpp:84
    assert np.all(g[cov][:, 0, 3] == 1) and np.all(g[cov][:, 1, 3] == 0) and np.all(g[cov][:, 2, 3] == 0)
    mrah = g[cov][:, 3]
    assert mrah[:, :3].min() >= 0 and mrah[:, :3].max() <= 1 and np.all(mrah[:, 3] == 0)   # no height map -> default 0
    n = np.linalg.norm(g[cov][:, 1, :3], axis=1)
    assert 0.95 < np.median(n) < 1.05 and n.max() < 1.5                        # TBN * (tex*2-1): near unit, not normalised
    p = g[cov][:, 0, :3]
    assert p[:, 1].min() >= -10.1 and p[:, 2].min() > 0                       # world positions: plane at y = -10, in front


def test_colour_differs_from_forward_only_by_the_binary16_stage():
    sc = _scene(configs.C3, (320, 180))
    fwd, fprim, fdepth, fst = bbo.render(sc)
    dfr, g, prim, depth, st = bbo.render_deferred(sc)
    assert st["n_prims"] == fst["n_prims"]
    # P*(V*p) against (P*V)*p moves clip positions by an ulp: coverage may differ on a handful of edge pixels at most
    assert (prim != fprim).mean() < 1e-3
    same = (prim == fprim) & (prim != bbo.NO_PRIM)
    d = np.abs(dfr[same][:, :3] - fwd[same][:, :3])
    rel = d / np.maximum(np.abs(fwd[same][:, :3]), 0.05)
    assert np.median(rel) < 2e-3 and rel.mean() < 0.01       # binary16 inputs: ~1e-3 relative, specular peaks more
    assert d.max() > 0                                       # but it IS a different image
    assert np.all(dfr[..., 3] == 1.0)                        # brdf.frag writes alpha 1 on every pixel, background too
    assert np.all(dfr[prim == bbo.NO_PRIM][:, :3] == 0)


def test_unmapped_normal_is_stored_unnormalised_and_a_light_at_the_origin_poisons_the_background():
    sc = _scene(configs.C2, (128, 72))
    sc.view["enable_normal_map"] = 0
    rgba, g, prim, _, _ = bbo.render_deferred(sc)
    cov = prim != bbo.NO_PRIM
    n = np.linalg.norm(g[cov][:, 1, :3].astype(np.float64), axis=1)
    assert n.max() <= 1.001 and np.median(n) > 0.99          # interpolated unit normals: length <= 1 (gbuffer.frag:29)
    sc.frame["lights"][0]["pos"] = (0, 0, 0)                 # 1/d^2 = inf at the cleared texel's position 0
    rgba, g, prim, _, _ = bbo.render_deferred(sc)
    assert np.isnan(rgba[prim == bbo.NO_PRIM][:, :3]).all()


def test_row_ranges_compose():
    sc = _scene(configs.C3, (200, 120))
    full, gf, pf, zf, _ = bbo.render_deferred(sc)
    a, ga, pa, za, _ = bbo.render_deferred(sc, 0, 50)
    b, gb, pb, zb, _ = bbo.render_deferred(sc, 50, 120)
    assert np.array_equal(np.concatenate([a[:50], b[50:]]).view(np.uint32), full.view(np.uint32))
    assert np.array_equal(np.concatenate([ga[:50], gb[50:]]), gf)
