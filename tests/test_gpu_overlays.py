"""Overlay subpass: light markers + corner gizmo over the presented image, GPU against the
oracle's bbo_overlay, byte for byte."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from bibim_renderer_amd import Renderer, BibimError, configs
from oracle import bbo, scenes

pytestmark = pytest.mark.gpu


def gizmo():
    g = np.load(os.path.join(GOLDEN, "gizmo.npz"))
    gv = np.zeros(len(g["vertices"]), bbo.GIZMO_VERTEX_DTYPE)
    gv["pos"], gv["color"], gv["normal"] = g["vertices"][:, 0:3], g["vertices"][:, 3:6], g["vertices"][:, 6:9]
    return g["vertices"], g["indices"], gv


def reference_image(sc, deferred, gv, gi, extent, enable=0, exposure=1.0):
    if deferred:
        hdr, _, _, depth, _ = bbo.render_deferred(sc)
    else:
        hdr, _, depth, _ = bbo.render(sc)
    base = bbo.present(hdr, enable, exposure)
    return base, bbo.overlay(sc.frame, sc.view, depth, base, gv, gi, extent)


@pytest.mark.parametrize("tile_mode", [0, 1])
@pytest.mark.parametrize("deferred", [False, True])
def test_markers_and_gizmo_match_the_oracle(maps64, tile_mode, deferred):
    raw, gi, gv = gizmo()
    cfg = configs.C3.scaled(640, 360, 64)
    sc = scenes.shaderball_scene(cfg, bbo.MaterialData(maps64))
    sc.frame["enable_tone_mapping"], sc.frame["exposure"] = 1, 1.3
    base, (want, st) = reference_image(sc, deferred, gv, gi, 100, 1, 1.3)
    assert (want != base).any(axis=2).sum() > 800            # four markers and the gizmo are really there
    r = Renderer(cfg.width, cfg.height)
    r.set_option("tile_mode", tile_mode)
    r.set_option("render_pass", int(deferred))
    r.set_option("overlays", 1)
    r.upload_gizmo(raw, gi)
    r.render_scene(sc)
    r.present()
    assert np.array_equal(r.read_presented(), base)
    r.draw_overlays(100)
    got = r.read_presented()
    bad = (got != want).any(axis=2)
    assert not bad.any(), (int(bad.sum()), np.argwhere(bad)[:5])
    r.close()


def test_markers_behind_geometry_close_to_the_camera_and_without_gizmo(maps64):
    """a marker inside a ball (hidden), one in front of everything, one crossing the near plane (clipped), one behind
    the camera; the gizmo switched off; then the gizmo alone in a smaller corner"""
    raw, gi, gv = gizmo()
    cfg = configs.C2.scaled(480, 270, 64)
    sc = scenes.shaderball_scene(cfg, bbo.MaterialData(maps64))
    sc.frame = scenes.frame_uniforms([scenes.light(0, pos=(0.0, -0.5, 2.0), color=(1, 0, 0), intensity=5.0),
                                      scenes.light(0, pos=(0.3, 0.1, 1.0), color=(0, 1, 0), intensity=5.0),
                                      scenes.light(0, pos=(0.02, 0.0, 0.15), color=(0.2, 0.3, 1), intensity=5.0),
                                      scenes.light(0, pos=(0.0, 0.0, -3.0), color=(1, 1, 0), intensity=5.0),
                                      scenes.light(2, dir=(0, -1, 0), color=(0.5, 0.5, 0.5), intensity=1.0)])
    hdr, _, depth, _ = bbo.render(sc)
    base = bbo.present(hdr, 0, 1.0)
    want, st = bbo.overlay(sc.frame, sc.view, depth, base, None, None, 0)
    assert st["n_clipped_prims"] > 0
    r = Renderer(cfg.width, cfg.height)
    r.set_option("overlays", 1)
    r.upload_gizmo(raw, gi)
    h = r.render_scene(sc)
    r.present()
    r.draw_overlays(0)
    assert np.array_equal(r.read_presented(), want)
    sc.frame["num_lights"] = 0                                    # no lights: only the gizmo, in a 64-pixel corner
    hdr, _, depth, _ = bbo.render(sc)
    base = bbo.present(hdr, 0, 1.0)
    want, _ = bbo.overlay(sc.frame, sc.view, depth, base, gv, gi, 64)
    r.render_scene(sc, h)
    r.present()
    r.draw_overlays(64)
    assert np.array_equal(r.read_presented(), want)
    r.close()


def test_gizmo_follows_the_camera_and_overflowing_bins_are_outgrown(maps64):
    raw, gi, gv = gizmo()
    cfg = configs.C3.scaled(400, 300, 64)
    sc = scenes.shaderball_scene(cfg, bbo.MaterialData(maps64))
    sc.view = scenes.view_uniforms((1.5, 3.0, -2.5), 40.0, -25.0, cfg.width, cfg.height, 1)
    hdr, _, depth, _ = bbo.render(sc)
    base = bbo.present(hdr, 0, 1.0)
    want, _ = bbo.overlay(sc.frame, sc.view, depth, base, gv, gi, 100)
    r = Renderer(cfg.width, cfg.height)
    r.set_option("overlays", 1)
    r.set_option("bin_cap", 8)                                    # far too small for 480 marker triangles in one tile
    r.upload_gizmo(raw, gi)
    r.render_scene(sc)
    r.present()
    r.draw_overlays(100)
    assert np.array_equal(r.read_presented(), want)
    r.close()


def test_error_paths(maps64):
    cfg = configs.C2.scaled(128, 96, 64)
    sc = scenes.shaderball_scene(cfg, bbo.MaterialData(maps64))
    r = Renderer(cfg.width, cfg.height)
    with pytest.raises(BibimError):
        r.draw_overlays(100)                                      # nothing rendered
    r.render_scene(sc)
    r.present()
    with pytest.raises(BibimError):
        r.draw_overlays(100)                                      # frame rendered without option "overlays"
    r.set_option("overlays", 1)
    r.render_scene(sc)
    with pytest.raises(BibimError):
        r.draw_overlays(100)                                      # not presented yet
    with pytest.raises(BibimError):
        r.upload_gizmo(np.zeros((3, 9), np.float32), np.array([0, 1, 7], np.uint32))   # index out of range
    r.close()


def test_frozen_fixture(maps64):
    """the committed overlaid image (reference default lights incl. a directional one, 48-pixel gizmo corner)"""
    z = np.load(os.path.join(GOLDEN, "overlays.npz"))
    raw, gi, gv = gizmo()
    sc = scenes.shaderball_scene(configs.C2.scaled(160, 90, 64), bbo.MaterialData(maps64))
    sc.frame = scenes.frame_uniforms(scenes.reference_default_lights(), 1, 1.7)
    r = Renderer(sc.width, sc.height)
    r.set_option("overlays", 1)
    r.upload_gizmo(raw, gi)
    r.render_scene(sc)
    r.present()
    assert np.array_equal(r.read_presented(), z["c2_160x90_base"])
    r.draw_overlays(48)
    assert np.array_equal(r.read_presented(), z["c2_160x90_overlaid"])
    r.close()
