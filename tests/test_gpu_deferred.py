"""Deferred variant on the GPU against the oracle's bbo_render_deferred: colour bits, winning
primitive, depth bits, G-buffer texels -- all exact."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from bibim_renderer_amd import BibimError, Renderer, configs, partition as P
from oracle import bbo, scenes

pytestmark = pytest.mark.gpu


def gpu_deferred(sc, tile_mode=None, gbuffer=True, **opts):
    r = Renderer(sc.width, sc.height)
    if tile_mode is not None:
        r.set_option("tile_mode", tile_mode)
    r.set_option("render_pass", 1)
    for k, v in opts.items():
        r.set_option(k, v)
    r.render_scene(sc)
    img = r.read_framebuffer()
    prim, depth = r.read_visibility()
    g = r.read_gbuffer() if gbuffer else None
    img2 = r.read_framebuffer()           # the G-buffer dump re-renders the frame: same bits
    st = r.stats()
    r.close()
    assert np.array_equal(img.view(np.uint32), img2.view(np.uint32))
    return img, g, prim, depth, st


def check(sc, tile_mode=None, **opts):
    ref, rg, rprim, rdepth, rst = bbo.render_deferred(sc)
    img, g, prim, depth, st = gpu_deferred(sc, tile_mode, **opts)
    assert np.array_equal(prim, rprim), f"{int((prim != rprim).sum())} pixels pick another primitive"
    assert np.array_equal(depth.view(np.uint32), rdepth.view(np.uint32))
    assert st["n_shaded"] == rst["n_shaded"] and st["n_clipped_prims"] == rst["n_clipped_prims"]
    assert np.array_equal(g.view(np.uint32), rg.view(np.uint32)), "G-buffer texels differ"
    assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)), "colour bits differ"
    return img, ref


@pytest.mark.parametrize("tile_mode", [0, 1])
def test_c2_and_c3_small(maps64, tile_mode, item_route_heavy):
    check(scenes.shaderball_scene(configs.C2.scaled(320, 180, 64), bbo.MaterialData(maps64)), tile_mode)
    check(scenes.shaderball_scene(configs.C3.scaled(640, 360, 64), bbo.MaterialData(maps64)), tile_mode)


def test_golden_fixture(maps64):
    z = np.load(os.path.join(GOLDEN, "deferred.npz"))
    sc = scenes.shaderball_scene(configs.C2.scaled(160, 90, 64), bbo.MaterialData(maps64))
    img, g, prim, depth, _ = gpu_deferred(sc)
    assert np.array_equal(img.view(np.uint32), z["c2_160x90_rgba_bits"])
    assert np.array_equal(g, z["c2_160x90_gbuffer_f16"].astype(np.float32))
    assert np.array_equal(prim, z["c2_160x90_prim"])


def test_unmapped_normal_default_material_and_nan_background(maps64):
    sc = scenes.shaderball_scene(configs.C2.scaled(256, 144, 64), bbo.MaterialData(maps64))
    sc.view["enable_normal_map"] = 0
    check(sc)
    sc.frame["lights"][0]["pos"] = (0, 0, 0)          # brdf.frag on the cleared texel: 1/d^2 = inf -> NaN background
    img, ref = check(sc)
    assert np.isnan(img[..., 0]).any()
    check(scenes.triangle_scene(96, 96))              # default material (1x1 maps, roughness 0), directional light


def test_mixed_map_sizes_with_a_height_map():
    rng = np.random.default_rng(5)
    maps = {"albedo": rng.integers(0, 256, (24, 40, 4), dtype=np.uint8), "roughness": rng.integers(40, 256, (16, 16, 4), dtype=np.uint8),
            "height": rng.integers(0, 256, (12, 20, 4), dtype=np.uint8), "normal": rng.integers(100, 156, (32, 32, 4), dtype=np.uint8)}
    sc = scenes.shaderball_scene(configs.C2.scaled(200, 120, 64), bbo.MaterialData(maps))
    _, rg, _, _, _ = bbo.render_deferred(sc)
    assert rg[..., 3, 3].max() > 0                    # the height channel is live
    check(sc)


def test_heavy_clipping_overflow_replay_and_frames_in_flight(maps64, item_route_heavy):
    cfg = configs.C3.scaled(384, 216, 64)
    sc = scenes.shaderball_scene(cfg, bbo.MaterialData(maps64))
    check(sc, bin_cap=8, frames_in_flight=3)
    sc.view = scenes.view_uniforms((-1.0, -0.55, 1.6), 35.0, -5.0, cfg.width, cfg.height, 1, near=0.05)  # camera between the balls
    check(sc)


def test_partition_and_present_on_the_deferred_image(maps64, item_route_heavy):
    cfg = configs.C3.scaled(512, 300, 64)
    sc = scenes.shaderball_scene(cfg, bbo.MaterialData(maps64))
    sc.frame["enable_tone_mapping"], sc.frame["exposure"] = 1, 1.4
    ref, _, _, _, rst = bbo.render_deferred(sc)
    world, band_rows = 4, 32
    shards, shards8, n = [], [], 0
    for rank in range(world):
        r = Renderer(cfg.width, cfg.height)
        r.set_option("render_pass", 1)
        r.set_partition(rank, world, band_rows)
        r.render_scene(sc)
        r.present()
        shards.append(r.read_shard())
        shards8.append(r.read_presented())
        n += r.stats()["n_shaded"]
        r.close()
    assert n == rst["n_shaded"]
    frame = P.unpack_gathered(np.stack(shards), cfg.height, band_rows)
    assert np.array_equal(frame.view(np.uint32), ref.view(np.uint32))
    assert np.array_equal(P.unpack_gathered(np.stack(shards8), cfg.height, band_rows), bbo.present(ref, 1, 1.4))


def test_switching_render_pass_between_frames(maps64):
    sc = scenes.shaderball_scene(configs.C2.scaled(192, 108, 64), bbo.MaterialData(maps64))
    fwd, _, _, _ = bbo.render(sc)
    dfr, _, _, _, _ = bbo.render_deferred(sc)
    r = Renderer(sc.width, sc.height)
    h = r.render_scene(sc)
    assert np.array_equal(r.read_framebuffer().view(np.uint32), fwd.view(np.uint32))
    r.set_option("render_pass", 1)
    h = r.render_scene(sc, h)
    assert np.array_equal(r.read_framebuffer().view(np.uint32), dfr.view(np.uint32))
    r.set_option("render_pass", 0)
    r.render_scene(sc, h)
    assert np.array_equal(r.read_framebuffer().view(np.uint32), fwd.view(np.uint32))
    with pytest.raises(Exception):
        r.read_gbuffer()                              # forward path has no G-buffer
    r.close()


@pytest.mark.parametrize("fused", [0, 1])
def test_gbuffer_views_show_the_attachments(maps64, fused):
    """buffer_visualize.frag (GBufferVisualizingOption, src/scene.h:27-35; recordCommand src/main.cpp:96-121): with the
    option on, the deferred frame is the rgb of one G-buffer attachment (binary16 values), alpha 1, cleared texels
    (0, 0, 0, 1); it is presented like any other frame"""
    sc = scenes.shaderball_scene(configs.C3.scaled(480, 270, 64), bbo.MaterialData(maps64))
    sc.frame["enable_tone_mapping"], sc.frame["exposure"] = 1, 0.8
    lit, gbuf, _, _, _ = bbo.render_deferred(sc)
    r = Renderer(sc.width, sc.height)
    r.set_option("render_pass", 1)
    r.set_option("present_fused", fused)
    h = None
    for view in (0, 1, 2, 3, -1, 2):
        r.set_option("gbuffer_view", view)
        h = r.render_scene(sc, h)
        want = lit if view < 0 else np.concatenate([gbuf[..., view, :3], np.ones(gbuf.shape[:2] + (1,), np.float32)], axis=-1)
        r.present()
        assert np.array_equal(r.read_presented(), bbo.present(want, 1, 0.8)), f"view {view}"
        if not fused:
            assert np.array_equal(r.read_framebuffer().view(np.uint32), want.view(np.uint32)), f"view {view}"
    with pytest.raises(BibimError):
        r.set_option("gbuffer_view", 4)          # MaterialIndex has no attachment behind it in the reference either
    r.set_option("render_pass", 0)               # forward path: the option is not consulted
    r.set_option("gbuffer_view", 1)
    r.render_scene(sc, h)
    if not fused:
        assert np.array_equal(r.read_framebuffer().view(np.uint32), bbo.render(sc)[0].view(np.uint32))
    r.close()


def test_full_size_c3_deferred_frame_against_the_literal_form():
    """The deferred pass at BASELINE's 4K size (C3: 16 balls, 4 lights, 2048^2 maps) against the oracle's LITERAL light loop
    -- brdf.frag:26-72 statement by statement on the binary16 G-buffer values -- not only against the contract form the
    kernel shares with the checker.  Tolerance, stated: absolute 1e-4 per channel (BASELINE.json)."""
    from bibim_renderer_amd import textures
    cfg = configs.C3
    sc = scenes.shaderball_scene(cfg, bbo.MaterialData(textures.make_material(cfg.texture_size)))
    literal, _, _, _, rst = bbo.render_deferred(sc, want_gbuffer=False, flags=bbo.FLAG_LITERAL)
    contract, _, _, _, _ = bbo.render_deferred(sc, want_gbuffer=False)
    r = Renderer(sc.width, sc.height)
    r.set_option("render_pass", 1)
    r.render_scene(sc)
    img = r.read_framebuffer()
    st = r.stats()
    r.close()
    assert st["n_shaded"] == rst["n_shaded"] and np.isfinite(literal).all() and np.isfinite(img).all()
    assert np.array_equal(img.view(np.uint32), contract.view(np.uint32)), "the deferred frame is no longer bit-exact at 4K"
    d = np.abs(img.astype(np.float64) - literal.astype(np.float64))
    assert d.max() <= 1e-4, float(d.max())
    assert np.array_equal(img[..., 3], literal[..., 3])
