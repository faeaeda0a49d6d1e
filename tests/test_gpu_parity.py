"""Parity tests proper: the HIP path, called through the C ABI (ctypes on libbibim_hip.so), against the CPU
oracle on the same seeded inputs.  Integer results (winning primitive, depth bits, coverage counts) must be
bit-exact; colour must satisfy BASELINE's tolerance |d| <= 1e-4 * max(1, |ref|) per channel (the contract is
written so that it is in fact bit-exact; the tests report when it is)."""
import json
from dataclasses import replace
import os

import numpy as np
import pytest

from conftest import GOLDEN, assert_frame_close
from bibim_renderer_amd import Renderer, BibimError, configs, textures
from oracle import bbo, scenes

pytestmark = pytest.mark.gpu


def gpu_render(scene, tile_mode=None, **opts):
    r = Renderer(scene.width, scene.height)
    if tile_mode is not None:
        r.set_option("tile_mode", tile_mode)  # default: the library's own (32x32)
    for k, v in opts.items():
        r.set_option(k, v)
    r.render_scene(scene)
    img = r.read_framebuffer()
    prim, depth = r.read_visibility()
    st = r.stats()
    r.close()
    return img, prim, depth, st


def check(scene, tile_mode=None, expect_exact=True, **opts):
    ref, rprim, rdepth, rst = bbo.render(scene)
    img, prim, depth, st = gpu_render(scene, tile_mode, **opts)
    assert np.array_equal(prim, rprim), f"{int((prim != rprim).sum())} pixels pick another primitive"
    assert np.array_equal(depth.view(np.uint32), rdepth.view(np.uint32))
    assert st["n_shaded"] == rst["n_shaded"] and st["n_prims"] == rst["n_prims"]
    assert st["n_clipped_prims"] == rst["n_clipped_prims"]
    assert_frame_close(img, ref)
    if expect_exact:
        assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)), "within tolerance but no longer bit-exact"
    return img, ref, st


@pytest.mark.parametrize("tile_mode", [0, 1])
def test_c2_small(maps64, tile_mode):
    check(scenes.shaderball_scene(configs.C2.scaled(320, 180, 64), bbo.MaterialData(maps64)), tile_mode)


@pytest.mark.parametrize("tile_mode", [0, 1])
@pytest.mark.parametrize("item_list", ["appended by k_raster (short frame)", "scanned by k_shade_items"])
def test_c3_small(maps256, tile_mode, item_list):
    """both ways a frame's item list comes about: a frame this small has its tiles append their items themselves; with
    no_tail_items = 0 it takes the long frames' route (k_shade_items + main and tail launch)"""
    opts = {} if item_list.startswith("appended") else {"no_tail_items": 0}
    check(scenes.shaderball_scene(configs.C3.scaled(960, 540, 256), bbo.MaterialData(maps256)), tile_mode, **opts)


def test_c5_small(maps64):
    check(scenes.shaderball_scene(configs.C5.scaled(1024, 576, 64), bbo.MaterialData(maps64)))


def test_golden_fixture_c2_160x90(maps64):
    """against the committed fixture (no oracle run)"""
    g = np.load(os.path.join(GOLDEN, "oracle_frames.npz"))
    img, prim, depth, _ = gpu_render(scenes.shaderball_scene(configs.C2.scaled(160, 90, 64), bbo.MaterialData(maps64)))
    assert np.array_equal(prim, g["c2_160x90_prim"])
    assert np.array_equal(depth.view(np.uint32), g["c2_160x90_depth_bits"])
    assert_frame_close(img, g["c2_160x90_rgba_bits"].view(np.float32))


def test_triangle_scene_default_material():
    g = np.load(os.path.join(GOLDEN, "oracle_frames.npz"))
    img, prim, depth, _ = gpu_render(scenes.triangle_scene(64, 64))
    assert np.array_equal(prim, g["triangle64_prim"])
    assert_frame_close(img, g["triangle64_rgba_bits"].view(np.float32))


def test_c2_full_size():
    mat = bbo.MaterialData(textures.make_material(2048))
    _, _, st = check(scenes.shaderball_scene(configs.C2, mat))
    assert st["n_shaded"] == json.load(open(os.path.join(GOLDEN, "n_shaded.json")))["c2"]["n_shaded"]


def test_c3_full_size_4k():
    """BASELINE's headline configuration at full size: 3840x2160, 16 balls, 4 lights, 2048^2 maps."""
    mat = bbo.MaterialData(textures.make_material(2048))
    _, _, st = check(scenes.shaderball_scene(configs.C3, mat))
    assert st["n_shaded"] == json.load(open(os.path.join(GOLDEN, "n_shaded.json")))["c3"]["n_shaded"]


@pytest.mark.parametrize("name", ["c2", "c3", "c5"])
def test_full_size_frame_against_the_literal_glsl_form(name):
    """The HIP frame against the oracle's LITERAL form -- forward_brdf.frag:29-70 / brdf.glsl statement by statement, no
    re-association -- at BASELINE's full sizes with 2048^2 maps.  `check` above is bit-exact against the CONTRACT form, which
    is the kernel's own evaluation order; THIS is the comparison that says the shipped order still renders the GLSL.
    Tolerance, stated: absolute 1e-4 per channel (BASELINE.json), hence also 1e-4 * max(1, |ref|)."""
    cfg = configs.CONFIGS[name]
    sc = scenes.shaderball_scene(cfg, bbo.MaterialData(textures.make_material(cfg.texture_size)))
    literal, n_shaded = bbo.render_bands(sc, flags=bbo.FLAG_LITERAL)
    r = Renderer(sc.width, sc.height)
    r.render_scene(sc)
    img = r.read_framebuffer()
    st = r.stats()
    r.close()
    assert st["n_shaded"] == n_shaded and np.isfinite(literal).all() and np.isfinite(img).all()
    d = np.abs(img.astype(np.float64) - literal.astype(np.float64))
    assert d.max() <= 1e-4, float(d.max())
    assert_frame_close(img, literal)
    assert np.array_equal(img[..., 3], literal[..., 3])


def _wall_scene(width, height, n_walls, maps, shrink=0.0, tilt=0.0):
    """n_walls screen-filling quads (two huge triangles each: every-tile-list entries) at different depths in front of the
    camera -- the case k_raster's light-tile path exists for, from one triangle per tile up to more entries than a wave
    has lanes.  shrink > 0 pulls the walls' edges into the frame, so that tiles see partial coverage too."""
    sc = scenes.triangle_scene(width, height, bbo.MaterialData(maps))
    v = np.zeros(4, bbo.VERTEX_DTYPE)
    e = 40.0 * (1.0 - shrink)
    v["pos"] = [(-e, -e, 0), (-e, e, 0), (e, e, 0), (e, -e, 0)]
    v["uv"] = [(0, 0), (0, 3), (3, 3), (3, 0)]
    v["normal"] = (0, 0, -1)
    v["tangent"] = (1, 0, 0)
    idx = np.array([0, 1, 2, 2, 3, 0], np.uint32)     # clockwise seen from the camera: front-facing
    inst = np.zeros(n_walls, bbo.INSTANCE_DTYPE)
    for i in range(n_walls):
        m = bbo.mat_mul(bbo.mat_translate(0.3 * i, -0.2 * i, 30.0 - 0.5 * i), bbo.mat_rotate_y(tilt * (i % 3 - 1)))
        inst[i] = scenes.instance(m)
    sc.draws = [bbo.DrawData(v, idx, inst, bbo.MaterialData(maps))]
    sc.frame = scenes.frame_uniforms([scenes.light(0, pos=(2, 3, 10), color=(1, 0.9, 0.8), intensity=400.0),
                                       scenes.light(2, dir=(0.2, -0.5, 1), color=(0.3, 0.4, 0.9), intensity=1.5)])
    return sc


@pytest.mark.parametrize("n_walls,shrink,tilt", [(1, 0.0, 0.0), (1, 0.7, 0.0), (2, 0.0, 12.0), (8, 0.75, 20.0), (33, 0.0, 0.0),
                                                 (40, 0.8, 7.0)])
def test_tiles_of_huge_triangles_through_the_light_tile_path(maps64, n_walls, shrink, tilt):
    """k_raster decides per wave, without LDS or a barrier, whether a tile with nothing binned is empty, covered by ONE
    every-tile-list triangle, or neither (then the general path): one wall (every interior tile full, the diagonal
    tiles see two triangles), walls that end inside the frame (empty tiles, partial tiles), several walls (never
    "one triangle"), and 66 / 80 list entries -- more than the 64 a wave classifies, so every tile takes the general path.
    Frame sizes off the tile grid on purpose."""
    for w, h in ((416, 240), (333, 207)):
        _, _, st = check(_wall_scene(w, h, n_walls, maps64, shrink, tilt))
        assert st["n_prims"] == 2 * n_walls


def test_normal_map_off_and_reference_default_lights(maps64):
    """reference defaults: EnableNormalMap = 0, three lights incl. a directional one and the radians-as-cosine quirk"""
    cfg = configs.C2.scaled(256, 144, 64)
    sc = scenes.shaderball_scene(cfg, bbo.MaterialData(maps64))
    sc.frame = scenes.frame_uniforms(scenes.reference_default_lights())
    sc.view["enable_normal_map"] = 0
    check(sc)


def test_spot_directional_and_unknown_light_types(maps64):
    cfg = configs.C2.scaled(200, 120, 64)
    sc = scenes.shaderball_scene(cfg, bbo.MaterialData(maps64))
    sc.frame = scenes.frame_uniforms([
        scenes.light(1, pos=(0.5, 2, 1.5), dir=(0, -1, 0.2), color=(1, 1, 0.9), intensity=40, inner=0.95, outer=0.7),
        scenes.light(2, dir=(-0.3, -1, 0.5), color=(0.2, 0.3, 0.9), intensity=2.0),
        scenes.light(5, pos=(0, 1, 0), color=(1, 1, 1), intensity=100.0),
        scenes.light(0, pos=(-1, 0.5, 1), color=(0.9, 0.2, 0.2), intensity=8.0)])
    check(sc)


def test_default_material_fallback_has_nans_in_the_same_places():
    """NULL maps => `default` material: roughness 0 makes brdf.glsl produce 0/0 at specular peaks; the GPU must
    put its NaNs exactly where the oracle does."""
    sc = scenes.shaderball_scene(configs.C2.scaled(240, 135, 16), bbo.MaterialData())
    img, ref, _ = check(sc, expect_exact=False)
    assert np.array_equal(np.isnan(img), np.isnan(ref))


def test_mixed_map_sizes_non_power_of_two_and_partial_material(item_route):
    rng = np.random.Generator(np.random.PCG64(11))
    maps = {"albedo": rng.integers(0, 256, (48, 80, 4), dtype=np.uint8),
            "roughness": rng.integers(60, 256, (17, 5, 4), dtype=np.uint8),
            "normal": textures.make_material(32)["normal"]}  # metallic / ao missing -> default maps
    check(scenes.shaderball_scene(configs.C2.scaled(224, 126, 32), bbo.MaterialData(maps)))


def test_two_materials_and_extra_indexed_draw(maps64):
    cfg = configs.C3.scaled(320, 180, 64)
    m1 = bbo.MaterialData(maps64)
    m2 = bbo.MaterialData({"albedo": textures.make_material(32, seed=7)["albedo"], "roughness": maps64["roughness"]})
    sc = scenes.shaderball_scene(cfg, m1)
    sc.draws[1].material = m2
    # a third draw: the plane mesh again, raised and tilted, sharing depth ties with nothing
    pv, pi = scenes.plane_mesh()
    inst = np.zeros(1, bbo.INSTANCE_DTYPE)
    inst[0] = scenes.instance(bbo.mat_mul(bbo.mat_translate(0.0, 0.5, 6.0), bbo.mat_mul(bbo.mat_rotate_x(35.0), bbo.mat_scale(3.0))))
    sc.draws.append(bbo.DrawData(pv, pi, inst, m2))
    check(sc)


def test_framebuffer_sizes_not_multiple_of_tile(maps64):
    # ... and very wide / very tall ones (aspect ratios of 500:1 either way, 16384 pixels on a side)
    for w, h in ((1, 1), (7, 3), (65, 64), (64, 65), (130, 33), (333, 211), (16384, 33), (31, 16384)):
        check(scenes.shaderball_scene(configs.C2.scaled(w, h, 64), bbo.MaterialData(maps64)))


def test_empty_frame_and_culled_geometry():
    fu = scenes.frame_uniforms([]); vu = scenes.view_uniforms((0, 0, 0), 0, 0, 96, 40, 0)
    r = Renderer(96, 40)
    r.set_frame_uniforms(fu); r.set_view_uniforms(vu)
    r.begin_frame(); r.end_frame()
    assert (r.read_framebuffer() == 0).all()
    assert r.stats()["n_shaded"] == 0
    r.close()
    v = np.zeros(6, bbo.VERTEX_DTYPE)
    v["pos"] = [(0, 0, 5), (0, 0, 5), (1, 1, 5), (0, 1, -5), (1, -1, -5), (-1, -1, -5)]  # zero area + behind the camera
    inst = np.zeros(1, bbo.INSTANCE_DTYPE); inst[0]["model"] = np.eye(4); inst[0]["inv_model"] = np.eye(4)
    sc = bbo.Scene(fu, vu, [bbo.DrawData(v, None, inst, bbo.MaterialData())], 96, 40)
    img, _, st = check(sc)
    assert (img == 0).all() and st["n_raster_tris"] == 0


def test_camera_inside_geometry_heavy_clipping(maps64, item_route_heavy):
    """camera between the balls, looking along the lattice: many primitives cross the near plane and the guard band"""
    cfg = configs.C3.scaled(256, 144, 64)
    sc = scenes.shaderball_scene(cfg, bbo.MaterialData(maps64))
    sc.view = scenes.view_uniforms((-1.0, -0.55, 1.6), 35.0, -5.0, cfg.width, cfg.height, 1, near=0.05)
    _, _, st = check(sc)
    assert st["n_clipped_prims"] >= 8


def hostile_scene(W=160, H=120, seed=0):
    """NaN, +-inf, huge, denormal and zero coordinates, uv, normals and instance matrices mixed into valid geometry"""
    rng = np.random.default_rng(seed)
    n_tri = 96
    v = np.zeros(n_tri * 3, bbo.VERTEX_DTYPE)
    pos = rng.uniform(-2, 2, (n_tri * 3, 3)).astype(np.float32)
    pos[:, 2] = rng.uniform(0.5, 8, n_tri * 3)
    bad = [np.nan, np.inf, -np.inf, 3e38, -3e38, 1e-42, 0.0]
    for t in range(0, n_tri, 3):
        pos[3 * t + rng.integers(0, 3), rng.integers(0, 3)] = bad[(t // 3) % len(bad)]
    v["pos"] = pos
    v["uv"] = rng.uniform(-3, 3, (n_tri * 3, 2))
    v["uv"][5] = (np.nan, np.inf)
    nrm = rng.standard_normal((n_tri * 3, 3)).astype(np.float32)
    nrm[7] = 0
    nrm[11] = np.nan
    v["normal"] = nrm
    v["tangent"] = rng.standard_normal((n_tri * 3, 3)).astype(np.float32)
    inst = np.zeros(3, bbo.INSTANCE_DTYPE)
    inst[0]["model"] = inst[0]["inv_model"] = np.eye(4, dtype=np.float32)
    m = np.eye(4, dtype=np.float32)
    m[3, :3] = (0.5, 0.2, 1.0)
    inst[1]["model"], inst[1]["inv_model"] = m, np.linalg.inv(m).astype(np.float32)
    m2 = np.eye(4, dtype=np.float32)
    m2[0, 0], m2[3, 2] = np.nan, np.inf
    inst[2]["model"] = inst[2]["inv_model"] = m2
    mat = bbo.MaterialData(textures.make_material(16))
    fu = scenes.frame_uniforms([scenes.light(0, pos=(0, 2, 0), color=(1, .8, .8), intensity=50.0),
                                scenes.light(2, dir=(0, 0, 0), color=(1, 1, 1), intensity=1.0)])   # normalize(0): NaN light
    vu = scenes.view_uniforms((0, 0, 0), 0.0, 0.0, W, H, 1)
    return bbo.Scene(fu, vu, [bbo.DrawData(v, None, inst, mat)], W, H, "hostile")


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_hostile_inputs_neither_hang_nor_differ(seed):
    """garbage in the vertex / instance streams must come out exactly as the oracle says (mostly: culled or clipped),
    on both render passes -- and must never fault or spin"""
    sc = hostile_scene(seed=seed)
    _, ref, st = check(sc)
    assert st["n_clipped_prims"] > 20
    dref, dg, dprim, ddepth, _ = bbo.render_deferred(sc)
    r = Renderer(sc.width, sc.height)
    r.set_option("render_pass", 1)
    r.render_scene(sc)
    img = r.read_framebuffer()
    g = r.read_gbuffer()
    r.close()
    assert np.array_equal(img.view(np.uint32), dref.view(np.uint32)) and np.array_equal(g.view(np.uint32), dg.view(np.uint32))


def test_depth_ties_follow_api_order():
    from test_oracle_kat import quad_scene
    for order in ("ab", "ba"):
        check(quad_scene(70, 50, 3.0, 3.0, order))
    check(quad_scene(70, 50, 2.0, 4.0, "ba"))


def test_bin_capacity_overflow_is_recovered(maps64, item_route_heavy):
    """tiny bins force the overflow path: the frame is re-rendered with larger bins, result unchanged"""
    sc = scenes.shaderball_scene(configs.C3.scaled(320, 180, 64), bbo.MaterialData(maps64))
    _, _, st = check(sc, bin_cap=8)
    assert st["bin_overflow"] >= 1


def test_broad_list_threshold_extremes(maps64):
    sc = scenes.shaderball_scene(configs.C3.scaled(400, 225, 64), bbo.MaterialData(maps64))
    check(sc, broad_threshold=1)       # everything touching more than one tile goes to the every-tile list
    check(sc, broad_threshold=100000)  # nothing does: the ground plane is binned into every tile it touches


@pytest.mark.parametrize("caps", [{"broad_cap": 1}, {"clip_cap": 1}, {"broad_cap": 2, "clip_cap": 2, "broad_threshold": 1}])
def test_every_tile_list_and_clip_arena_overflow_through_a_clipped_primitive(maps64, caps, item_route_heavy):
    """The clip path reserves a RUN of list entries / arena slots per primitive and writes none of them when the run
    does not fit: the overflowed frame must not consume the unwritten part (it takes no every-tile entry at all), and the
    frame rendered after the growth is the oracle's.  Once with a synchronising call right after the first frame, once
    streaming frames with no synchronising call in between (the capacities then grow from the pinned overflow words)."""
    sc = scenes.shaderball_scene(configs.C3.scaled(400, 225, 64), bbo.MaterialData(maps64))
    ref, rprim, _, rst = bbo.render(sc)
    assert rst["n_clipped_prims"] > 0, "the scene must send a primitive through the clip path"
    _, _, st = check(sc, **caps)
    assert st["bin_overflow"] >= 1
    r = Renderer(sc.width, sc.height)
    for k, v in caps.items():
        r.set_option(k, v)
    h = None
    for _ in range(12):  # no synchronising call: overflowed frames are incomplete, later ones must be right
        h = r.render_scene(sc, h)
    img = r.read_framebuffer()
    prim, _ = r.read_visibility()
    r.close()
    assert np.array_equal(prim, rprim)
    assert np.array_equal(img.view(np.uint32), ref.view(np.uint32))


@pytest.mark.parametrize("frames_in_flight", [1, 2, 3])
def test_frames_in_flight_streams_of_different_frames(maps64, frames_in_flight, item_route):
    """two frames in flight on two streams: alternating scenes back to back, no synchronisation in between,
    every frame must still come out exactly as when rendered alone"""
    sa = scenes.shaderball_scene(configs.C3.scaled(448, 252, 64), bbo.MaterialData(maps64))
    sb = scenes.shaderball_scene(configs.C2.scaled(448, 252, 64), sa.draws[0].material)
    ra, _, _, _ = bbo.render(sa)
    rb, _, _, _ = bbo.render(sb)
    r = Renderer(sa.width, sa.height)
    r.set_option("frames_in_flight", frames_in_flight)
    h = None
    for i in range(9):
        h = r.render_scene(sa if i % 2 == 0 else sb, h)
    last = r.read_framebuffer()            # frame 8 = scene a
    assert np.array_equal(last.view(np.uint32), ra.view(np.uint32))
    h = r.render_scene(sb, h)
    assert np.array_equal(r.read_framebuffer().view(np.uint32), rb.view(np.uint32))
    # single external output buffer shared by consecutive frames: raster of N+1 must wait for shade of N
    import torch
    out = torch.zeros((sa.height, sa.width, 4), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()   # torch fills on its own stream; the library does not wait for that one
    r.set_output_device_ptr(out.data_ptr(), out.numel() * 4)
    for i in range(6):
        h = r.render_scene(sb if i % 2 == 0 else sa, h)
    r.synchronize()
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy().view(np.uint32), ra.view(np.uint32))
    r.close()


def test_replay_is_idempotent_and_frames_are_independent(maps64):
    sc = scenes.shaderball_scene(configs.C3.scaled(384, 216, 64), bbo.MaterialData(maps64))
    r = Renderer(sc.width, sc.height)
    h = r.render_scene(sc)
    a = r.read_framebuffer()
    for _ in range(3):
        r.replay_frame()
    b = r.read_framebuffer()
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    # a different frame on the same context, then the first one again
    sc2 = scenes.shaderball_scene(configs.C2.scaled(384, 216, 64), sc.draws[0].material)
    r.render_scene(sc2, h)
    c = r.read_framebuffer()
    ref2, _, _, _ = bbo.render(sc2)
    assert_frame_close(c, ref2)
    r.render_scene(sc, h)
    assert np.array_equal(r.read_framebuffer().view(np.uint32), a.view(np.uint32))
    r.close()


def test_error_codes():
    r = Renderer(32, 32)
    with pytest.raises(BibimError) as e:
        r.draw(0, 0, np.zeros(32, np.float32))          # outside begin/end
    assert e.value.code == -6
    r.begin_frame()
    with pytest.raises(BibimError) as e:
        r.draw(3, 0, np.zeros(32, np.float32))          # bad mesh handle
    assert e.value.code == -5
    fu = np.zeros((), bbo.FRAME_DTYPE); fu["num_lights"] = 100
    with pytest.raises(BibimError) as e:
        r.set_frame_uniforms(fu)                         # reference asserts NumLights < 100
    assert e.value.code == -1
    v = np.zeros(3, bbo.VERTEX_DTYPE)
    with pytest.raises(BibimError):
        r.upload_mesh(v, np.array([0, 1, 3], np.uint32))  # index out of range
    r.close()
    with pytest.raises(BibimError):
        Renderer(0, 10)


def test_reciprocal_of_the_contract_is_the_ieee_division_for_every_float():
    """bb_rcp = v_rcp_f32 + one Newton step on the GPU, 1.0f / x in the oracle: identical for all 2^31 magnitudes and
    both signs (denormals, zero, inf and NaN take the IEEE path on both sides)"""
    r = Renderer(64, 64)
    assert r.selftest_rcp(0x00000000, 0x7FFFFFFF) == 0
    assert r.selftest_rcp(0x7F800000, 0x7F800010) == 0 and r.selftest_rcp(0, 0x00800010) == 0
    r.close()


def test_destroying_the_context_before_its_scene_is_harmless():
    """garbage-collected hosts may drop the context first: the scene's late bbr_free_mesh calls must fail cleanly"""
    import ctypes as C
    from bibim_renderer_amd import _capi
    L = _capi.lib()
    ctx = C.c_void_p()
    assert L.bbr_create(64, 64, 0, C.byref(ctx)) == 0
    scene = L.bbs_triangle_scene_create(ctx)
    assert scene
    assert L.bbr_destroy(ctx) == 0
    assert L.bbr_destroy(ctx) != 0            # already gone: BBR_ERR_BAD_HANDLE, not a double free
    assert L.bbr_free_mesh(ctx, 0) != 0
    L.bbs_scene_destroy(C.c_void_p(scene))    # frees its mesh through the dead context: must not crash


def test_tone_map_next_row(maps64):
    sc = scenes.shaderball_scene(configs.C2.scaled(160, 90, 64), bbo.MaterialData(maps64))
    r = Renderer(sc.width, sc.height)
    r.render_scene(sc)
    hdr = r.read_framebuffer()
    r.tone_map(1, 1.7)
    ldr = r.read_framebuffer()
    r.close()
    want = bbo.tone_map(hdr, 1, 1.7)
    assert np.array_equal(ldr.view(np.uint32), want.view(np.uint32))  # exp is a fixed sequence shared with the oracle


@pytest.mark.parametrize("enable,exposure,hdr16", [(0, 1.0, 1), (1, 1.7, 1), (1, 0.6, 0)])
def test_present_bytes_match_the_oracle(maps64, enable, exposure, hdr16):
    """the step after the path: binary16 HDR attachment -> tone map -> sRGB UNORM8, byte for byte"""
    sc = scenes.shaderball_scene(configs.C3.scaled(640, 360, 64), bbo.MaterialData(maps64))
    sc.frame["enable_tone_mapping"], sc.frame["exposure"] = enable, exposure
    ref, _, _, _ = bbo.render(sc)
    r = Renderer(sc.width, sc.height)
    r.render_scene(sc)
    r.present(hdr16=hdr16)
    got = r.read_presented()
    hdr = r.read_framebuffer()                      # present leaves the fp32 frame untouched
    r.close()
    assert np.array_equal(hdr.view(np.uint32), ref.view(np.uint32))
    want = bbo.present(ref, enable, exposure, hdr16)
    assert got.shape == want.shape and np.array_equal(got, want)
    assert np.all(got[..., 3] == 255) and got[..., :3].max() > 100


def test_present_golden_fixture_and_special_values(maps64):
    """committed presented bytes of the golden C2 160x90 frame; NaN / inf pixels of the default-material frame"""
    z = np.load(os.path.join(GOLDEN, "presented.npz"))
    sc = scenes.shaderball_scene(configs.C2.scaled(160, 90, 64), bbo.MaterialData(maps64))
    sc.frame["enable_tone_mapping"], sc.frame["exposure"] = 1, 1.7
    r = Renderer(sc.width, sc.height)
    r.render_scene(sc)
    r.present()
    assert np.array_equal(r.read_presented(), z["c2_160x90_tonemapped"])
    r.close()
    tri = scenes.triangle_scene(96, 96)             # default material: roughness 0 -> NaN where N.H = 1
    ref, _, _, _ = bbo.render(tri)
    r = Renderer(96, 96)
    r.render_scene(tri)
    r.present()
    assert np.array_equal(r.read_presented(), bbo.present(ref, 0, 1.0))
    r.close()


def test_present_buffer_on_every_rounding_boundary():
    """k_present reads the sRGB byte from a table keyed by the float's top bits instead of searching the thresholds:
    same byte for every float -- all thresholds and all table cell edges +-3 ulp, the binary16 rounding boundaries,
    specials, and 4M random bit patterns; with and without the binary16 stage and the tone map"""
    import torch
    thr = bbo.srgb_thresholds().view(np.uint32).astype(np.int64)
    cells = ((np.arange(13 * 256 + 1, dtype=np.int64) + (114 << 8)) << 15)
    halves = np.arange(0, 0x7C00, dtype=np.uint16).view(np.float16).astype(np.float32)   # every non-negative binary16
    mids = ((halves[:-1].astype(np.float64) + halves[1:].astype(np.float64)) / 2).astype(np.float32).view(np.uint32).astype(np.int64)
    near = np.concatenate([b + d for b in (thr, cells, mids) for d in range(-3, 4)])
    rng = np.random.default_rng(11)
    specials = np.array([0, 0x80000000, 0x7F800000, 0xFF800000, 0x7FC00000, 0xFFC00000, 0x7F800001, 1, 0x007FFFFF, 0x00800000,
                         0x3F800000, 0x3F7FFFFF, 0x3F800001, 0x477FE000, 0x477FF000, 0x477FEFFF, 0x38800000, 0x387FFFFF], np.int64)
    bits = np.concatenate([near, specials, rng.integers(0, 1 << 32, 1 << 22, dtype=np.int64),
                           rng.integers(0x38000000, 0x40000000, 1 << 21, dtype=np.int64)]).astype(np.uint32)
    bits = np.resize(bits, (bits.size + 3) // 4 * 4)
    src = bits.view(np.float32).reshape(-1, 4).copy()
    d_src = torch.from_numpy(src).cuda()
    d_out = torch.zeros(src.shape, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()   # torch fills on its own stream; the library does not wait for that one
    r = Renderer(64, 64)
    for enable, exposure, hdr16 in ((0, 1.0, 0), (0, 1.0, 1), (1, 0.9, 1), (1, 2.5, 0)):
        r.present_buffer(d_src.data_ptr(), d_out.data_ptr(), src.shape[0], enable, exposure, hdr16)
        r.synchronize()
        torch.cuda.synchronize()
        got = d_out.cpu().numpy()
        want = bbo.present(src, enable, exposure, hdr16)
        bad = np.nonzero((got != want).any(axis=1))[0]
        assert bad.size == 0, (enable, exposure, hdr16, src[bad[:4]], got[bad[:4]], want[bad[:4]])
    r.close()


def test_present_after_an_overflow_replay_and_into_a_caller_buffer(maps64, item_route):
    import torch
    sc = scenes.shaderball_scene(configs.C3.scaled(320, 180, 64), bbo.MaterialData(maps64))
    sc.frame["enable_tone_mapping"], sc.frame["exposure"] = 1, 1.0
    ref, _, _, _ = bbo.render(sc)
    want = bbo.present(ref, 1, 1.0)
    r = Renderer(sc.width, sc.height)
    r.set_option("bin_cap", 8)                      # the frame overflows its bins and is rendered again on sync
    out = torch.zeros((sc.height, sc.width, 4), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()   # torch fills on its own stream; the library does not wait for that one
    r.render_scene(sc)
    r.present(out.data_ptr())
    got = r.read_presented()                        # synchronises, replays, presents again
    assert r.stats()["bin_overflow"] >= 1
    torch.cuda.synchronize()
    assert np.array_equal(got, want) and np.array_equal(out.cpu().numpy(), want)
    r.close()


def test_diagnostic_reads_keep_the_presented_image(maps64):
    """bbr_read_visibility / bbr_read_gbuffer render the frame once more: the slot's presentation state must survive it
    (bbr_read_presented afterwards, and a caller's presented buffer refreshed with the same pixels)."""
    import torch
    sc = scenes.shaderball_scene(configs.C2.scaled(256, 144, 64), bbo.MaterialData(maps64))
    sc.frame["enable_tone_mapping"], sc.frame["exposure"] = 1, 1.5
    ref, rprim, _, _ = bbo.render(sc)
    want = bbo.present(ref, 1, 1.5)
    for deferred in (0, 1):
        r = Renderer(sc.width, sc.height)
        r.set_option("render_pass", deferred)
        out = torch.zeros((sc.height, sc.width, 4), dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        r.render_scene(sc)
        r.present(out.data_ptr())
        prim, _ = r.read_visibility()
        if deferred:
            r.read_gbuffer()
        else:
            assert np.array_equal(prim, rprim)
            assert np.array_equal(r.read_presented(), want)
            torch.cuda.synchronize()
            assert np.array_equal(out.cpu().numpy(), want)
        first = r.read_presented()
        assert np.array_equal(first, r.read_presented())
        r.close()


def test_api_lifecycle_user_stream_frees_and_timing(maps64):
    """the less-travelled entry points: rendering on the caller's stream, freeing meshes / materials between frames
    (and the stale-handle errors afterwards), timing queries, re-creating a context at another size"""
    import torch
    sc = scenes.shaderball_scene(configs.C2.scaled(256, 144, 64), bbo.MaterialData(maps64))
    ref, _, _, _ = bbo.render(sc)
    r = Renderer(sc.width, sc.height)
    # (1) everything on a caller-owned stream, ordered with the caller's own work on it
    stream = torch.cuda.Stream()
    out = torch.full((sc.height, sc.width, 4), 7.0, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()   # torch fills on its own stream; the library does not wait for that one
    stream.wait_stream(torch.cuda.current_stream())   # the fill above ran on torch's current stream
    r.set_stream(stream.cuda_stream)
    r.set_output_device_ptr(out.data_ptr(), out.numel() * 4)
    h = r.render_scene(sc)
    r.synchronize()                             # first frame of a scene: sizes the bins (re-renders on overflow)
    with torch.cuda.stream(stream):
        out.fill_(7.0)
        h = r.render_scene(sc, h)
        doubled = out * 2                       # queued behind the frame on the same stream
    stream.synchronize()
    assert np.array_equal(out.cpu().numpy().view(np.uint32), ref.view(np.uint32))
    assert np.array_equal(doubled.cpu().numpy(), ref * 2)
    r.set_stream(None)
    r.set_output_device_ptr(None, 0)
    # (2) timing queries
    r.set_option("timing", 1)
    r.timing_reset()
    h = r.render_scene(sc, h)
    frame_ms, shade_ms = r.last_frame_time_ms()
    n, f, g, ra, s = r.timing_summary()
    assert n == 1 and 0 < shade_ms <= frame_ms and f > 0 and g > 0 and ra > 0 and s > 0
    r.set_option("timing", 0)
    assert np.array_equal(r.read_framebuffer().view(np.uint32), ref.view(np.uint32))
    # (3) freeing resources: handles die, other resources keep working, slots are not recycled into live handles
    meshes, mats = list(h["mesh"].values()), list(h["mat"].values())
    r.free_mesh(meshes[0])
    with pytest.raises(BibimError):
        r.free_mesh(meshes[0])
    r.begin_frame()
    with pytest.raises(BibimError):
        r.draw(meshes[0], mats[0], sc.draws[0].instances)
    r.draw(meshes[1], mats[0], sc.draws[1].instances)   # the plane alone still renders
    r.end_frame()
    plane_only = bbo.Scene(sc.frame, sc.view, [sc.draws[1]], sc.width, sc.height, "plane")
    pref, _, _, _ = bbo.render(plane_only)
    assert np.array_equal(r.read_framebuffer().view(np.uint32), pref.view(np.uint32))
    r.free_material(mats[0])
    r.begin_frame()
    with pytest.raises(BibimError):
        r.draw(meshes[1], mats[0], sc.draws[1].instances)
    r.end_frame()
    r.close()
    # (4) "resize" = destroy + create (src/main.cpp:1042-1070 recreates the swapchain-dependent objects)
    sc2 = scenes.shaderball_scene(configs.C2.scaled(200, 300, 64), bbo.MaterialData(maps64))
    r2 = Renderer(sc2.width, sc2.height)
    r2.render_scene(sc2)
    ref2, _, _, _ = bbo.render(sc2)
    assert np.array_equal(r2.read_framebuffer().view(np.uint32), ref2.view(np.uint32))
    r2.close()


@pytest.mark.parametrize("deferred", [0, 1])
@pytest.mark.parametrize("enable,exposure", [(0, 1.0), (1, 1.6)])
def test_fused_presentation_writes_the_same_bytes(maps64, deferred, enable, exposure, item_route):
    """option present_fused: the raster / shade kernels produce the presented RGBA8 image themselves (no fp32 frame, no
    k_present): byte-identical to rendering + bbr_present, i.e. to the oracle's bbo_present of the oracle's frame"""
    import torch
    sc = scenes.shaderball_scene(configs.C3.scaled(640, 360, 64), bbo.MaterialData(maps64))
    sc.frame["enable_tone_mapping"], sc.frame["exposure"] = enable, exposure
    ref = bbo.render_deferred(sc)[0] if deferred else bbo.render(sc)[0]
    want = bbo.present(ref, enable, exposure)
    r = Renderer(sc.width, sc.height)
    r.set_option("render_pass", deferred)
    r.set_option("present_fused", 1)
    r.set_option("bin_cap", 16)                          # the first frame overflows and is re-rendered on the read
    out = torch.zeros((sc.height, sc.width, 4), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()   # torch fills on its own stream; the library does not wait for that one
    h = r.render_scene(sc)
    r.present(out.data_ptr())                            # fused: a copy of the image into the caller's buffer
    got = r.read_presented()
    assert np.array_equal(got, want)
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), want)
    with pytest.raises(BibimError):
        r.read_framebuffer()                             # there is no fp32 frame in this mode
    r.set_option("present_fused", 0)                     # and back: the fp32 frame returns
    r.render_scene(sc, h)
    assert np.array_equal(r.read_framebuffer().view(np.uint32), ref.view(np.uint32))
    r.close()


def test_fused_presentation_special_values_partition_and_overlays(maps64, item_route):
    from bibim_renderer_amd import partition as P
    tri = scenes.triangle_scene(96, 96)                  # default material: NaN where N.H = 1
    ref, _, _, _ = bbo.render(tri)
    r = Renderer(96, 96)
    r.set_option("present_fused", 1)
    r.render_scene(tri)
    assert np.array_equal(r.read_presented(), bbo.present(ref, 0, 1.0))
    r.close()
    cfg = configs.C3.scaled(512, 300, 64)
    sc = scenes.shaderball_scene(cfg, bbo.MaterialData(maps64))
    sc.frame["enable_tone_mapping"], sc.frame["exposure"] = 1, 1.2
    ref, _, depth, _ = bbo.render(sc)
    want = bbo.present(ref, 1, 1.2)
    shards = []
    for rank in range(3):
        r = Renderer(cfg.width, cfg.height)
        r.set_option("present_fused", 1)
        r.set_partition(rank, 3, 32)
        r.render_scene(sc)
        shards.append(r.read_presented())
        r.close()
    assert np.array_equal(P.unpack_gathered(np.stack(shards), cfg.height, 32), want)
    g = np.load(os.path.join(GOLDEN, "gizmo.npz"))
    gv = np.zeros(len(g["vertices"]), bbo.GIZMO_VERTEX_DTYPE)
    gv["pos"], gv["color"], gv["normal"] = g["vertices"][:, 0:3], g["vertices"][:, 3:6], g["vertices"][:, 6:9]
    over, _ = bbo.overlay(sc.frame, sc.view, depth, want, gv, g["indices"], 100)
    r = Renderer(cfg.width, cfg.height)
    r.set_option("present_fused", 1)
    r.set_option("overlays", 1)
    r.upload_gizmo(g["vertices"], g["indices"])
    r.render_scene(sc)
    r.present()
    r.draw_overlays(100)
    assert np.array_equal(r.read_presented(), over)
    r.close()


def test_resize_keeps_resources_and_renders_the_new_extent_exactly(maps64):
    """onWindowResize (src/main.cpp:1042-1061): same context, meshes and materials; bigger, smaller, off the tile grid,
    with three frames in flight, presentation and the deferred pass switched on along the way"""
    mat = bbo.MaterialData(maps64)
    r = Renderer(320, 180)
    r.set_option("frames_in_flight", 3)
    handles, deferred, keep = None, 0, []
    for (w, h, cfg, extra) in [(320, 180, configs.C3, {}), (701, 397, configs.C3, {}), (96, 50, configs.C2, {}),
                               (640, 360, configs.C5, {"render_pass": 1}), (320, 180, configs.C3, {"render_pass": 0})]:
        if (w, h) != (r.width, r.height):
            r.resize(w, h)
            with pytest.raises(BibimError) as e:
                r.read_framebuffer()                     # nothing rendered at the new extent yet
            assert e.value.code == -6
        for k, v in extra.items():
            r.set_option(k, v)
        deferred = extra.get("render_pass", deferred)
        sc = scenes.shaderball_scene(cfg.scaled(w, h, 64), mat)
        keep.append(sc)                                  # (handles are keyed by object identity)
        for _ in range(4):                               # every slot gets its buffers back
            handles = r.render_scene(sc, handles)
        img = r.read_framebuffer()
        prim, depth = r.read_visibility()
        if deferred == 1:
            ref, _, rprim, rdepth, _ = bbo.render_deferred(sc, want_gbuffer=False)
        else:
            ref, rprim, rdepth, _ = bbo.render(sc)
        assert img.shape == (h, w, 4)
        assert np.array_equal(prim, rprim) and np.array_equal(depth.view(np.uint32), rdepth.view(np.uint32))
        assert np.array_equal(img.view(np.uint32), ref.view(np.uint32))
        r.present()
        assert np.array_equal(r.read_presented(), bbo.present(ref, int(sc.frame["enable_tone_mapping"]), float(sc.frame["exposure"])))
    assert len(handles["mat"]) == 1 and len(handles["mesh"]) == 1 + len(keep)   # the ball and the maps went up once
    r.begin_frame()
    with pytest.raises(BibimError):
        r.resize(64, 64)                                 # not inside a frame
    r.end_frame()
    with pytest.raises(BibimError):
        r.resize(0, 64)
    r.resize(320, 180)                                   # same extent: a no-op apart from the wait
    r.close()


@pytest.mark.parametrize("mode", [0, 1, 2, "switching"])
def test_stream_layouts_render_the_same_frames(maps64, mode):
    """option "stream_layout": stage streams, k_raster on its own stream, one stream per frame slot (the default), or
    switched by the caller every few frames with frames in flight -- the frames, their presented images and tone-mapped
    copies are the oracle's, bit for bit, throughout"""
    sc = scenes.shaderball_scene(configs.C3.scaled(640, 360, 64), bbo.MaterialData(maps64))
    sc.frame["enable_tone_mapping"], sc.frame["exposure"] = 1, 1.3
    ref, _, _, _ = bbo.render(sc)
    want8 = bbo.present(ref, 1, 1.3)
    r = Renderer(sc.width, sc.height)
    r.set_option("frames_in_flight", 3)
    assert r.stream_layout_state() == (2, 1, [0, 0, 0])      # the documented default; nothing is timed at run time
    if mode != "switching":
        r.set_option("stream_layout", mode)
    h = None
    for i in range(260):
        if mode == "switching" and i % 7 == 0:
            r.set_option("stream_layout", (i // 7) % 3)
        h = r.render_scene(sc, h)
        if i % 5 == 0:
            r.present()                  # queued behind the frame's k_shade, whichever stream that ran on
        if i % 23 == 0 or i == 259:      # the read-back drains the pipeline: the switches happen at different depths
            assert np.array_equal(r.read_framebuffer().view(np.uint32), ref.view(np.uint32)), f"frame {i}"
            if i % 5 == 0:
                assert np.array_equal(r.read_presented(), want8), f"presented frame {i}"
    layout, decided, ms = r.stream_layout_state()
    assert decided and ms == [0, 0, 0] and (layout == mode or mode == "switching")
    sc2 = scenes.shaderball_scene(configs.C2.scaled(640, 360, 64), sc.draws[0].material)   # another workload, more fragments
    r.render_scene(sc2, h)                                                                  # than the slots expect
    ref2, _, _, _ = bbo.render(sc2)
    assert np.array_equal(r.read_framebuffer().view(np.uint32), ref2.view(np.uint32))
    r.tone_map(1, 1.3)                   # in place, on the stream the frame's k_shade ran on
    with pytest.raises(BibimError):
        r.set_option("stream_layout", 3)
    with pytest.raises(BibimError):
        r.set_option("stream_layout", -1)
    with pytest.raises(BibimError):
        r.set_option("heavy_tiles", -2)      # (-1 = automatic, 0 = off, n = a bin's reference count)
    r.set_option("heavy_tiles", -1)
    r.close()


@pytest.mark.parametrize("heavy", [0, 4])
@pytest.mark.parametrize("layout", [0, 2])
def test_a_frame_with_many_more_fragments_than_the_one_before(maps64, layout, heavy):
    """k_shade's main launch is sized from the item count of the frame the slot rendered before; whatever lies behind it
    is shaded by the small persistent tail launch.  A nearly empty frame followed by a full one (and back) in every
    frame slot, with frames in flight: every frame is the oracle's, bit for bit.
    heavy = 4: k_raster's heavy rows are sized from the previous frame of the slot too -- the first big frame's list does
    not fit the rows the small frame left (it is rasterised in plain screen order), the next one's does, and the small
    frame that follows has far more rows than entries."""
    big = scenes.shaderball_scene(configs.C3.scaled(960, 540, 64), bbo.MaterialData(maps64))
    small = scenes.triangle_scene(960, 540)
    ref_big, _, _, st_big = bbo.render(big)
    ref_small, _, _, st_small = bbo.render(small)
    assert st_big["n_shaded"] > 10 * st_small["n_shaded"] > 0
    r = Renderer(960, 540)
    r.set_option("frames_in_flight", 3)
    r.set_option("stream_layout", layout)
    r.set_option("no_tail_items", 0)             # (a frame this small would otherwise be launched at full coverage, without a tail)
    r.set_option("heavy_tiles", heavy)
    hb = hs = None
    for rep in range(3):
        for _ in range(4):                       # every slot has seen the small frame
            hs = r.render_scene(small, hs)
        assert np.array_equal(r.read_framebuffer().view(np.uint32), ref_small.view(np.uint32))
        for i in range(4):                       # ... and now gets the big one: the tail shades nearly all of it
            hb = r.render_scene(big, hb)
            if i in (0, 3):
                assert np.array_equal(r.read_framebuffer().view(np.uint32), ref_big.view(np.uint32)), (rep, i)
        assert r.stats()["n_shaded"] == st_big["n_shaded"]
    r.close()


@pytest.mark.parametrize("layout", [0, 1, 2])
def test_caller_buffers_shared_by_frames_in_flight(maps64, layout):
    """three frames in flight into two caller-owned buffers, no synchronisation in between: a slow frame into A, an empty
    one into B, an empty one into A.  The last frame's clear must land after the slow frame's shading although the frame
    right before it wrote elsewhere: "behind the previous frame" is not enough"""
    import torch
    from dataclasses import replace
    cfg = replace(configs.C5.scaled(1920, 1080, 64), lights=(configs.C5.lights * 4)[:32])     # slow to shade
    heavy = scenes.shaderball_scene(cfg, bbo.MaterialData(maps64))
    r = Renderer(cfg.width, cfg.height)
    r.set_option("frames_in_flight", 3)
    r.set_option("stream_layout", layout)
    h = r.render_scene(heavy)
    r.synchronize()                                   # capacities sized
    out_a = torch.full((cfg.height, cfg.width, 4), 7.0, dtype=torch.float32, device="cuda")
    out_b = torch.full((cfg.height, cfg.width, 4), 7.0, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()   # torch fills on its own stream; the library does not wait for that one
    empty_f, empty_v = scenes.frame_uniforms([]), heavy.view
    def target(t):
        r.set_output_device_ptr(t.data_ptr(), t.numel() * 4)
    def empty_frame():
        r.set_frame_uniforms(empty_f); r.set_view_uniforms(empty_v)
        r.begin_frame(); r.end_frame()
    for rep in range(6):                              # four frames a round: the slow one visits every slot (and priority)
        target(out_b); empty_frame()
        target(out_a); h = r.render_scene(heavy, h)
        target(out_b); empty_frame()                  # the frame before the last one writes elsewhere ...
        target(out_a); empty_frame()                  # ... and the last one must still wait for the slow one
        r.synchronize()
        torch.cuda.synchronize()
        assert int((out_a != 0).sum().item()) == 0, f"round {rep}: pixels of an earlier frame survived the last frame's clear"
        assert int((out_b != 0).sum().item()) == 0
    r.close()


def test_ninety_nine_lights_of_every_type_and_many_draws(maps64):
    """the most the reference allows (NumLights < 100, src/main.cpp:1289) with point, spot and directional lights mixed
    (and a few of an unknown type, which the shader skips), and the scene split into many small draws"""
    rng = np.random.Generator(np.random.PCG64(99))
    base = scenes.shaderball_scene(configs.C3.scaled(320, 180, 64), bbo.MaterialData(maps64))
    lights = []
    for i in range(99):
        kind = (0, 1, 2, 0, 1, 2, 0, 7)[i % 8]
        lights.append(scenes.light(type=kind, pos=tuple(rng.uniform(-4, 4, 3) + (0, 3, 3)), dir=tuple(rng.uniform(-1, 1, 3)),
                                   color=tuple(rng.uniform(0.2, 1.0, 3)), intensity=float(rng.uniform(0.5, 6.0)),
                                   inner=float(rng.uniform(0.7, 0.95)), outer=float(rng.uniform(0.3, 0.7))))
    frame = scenes.frame_uniforms(lights)
    draws = []
    ball = base.draws[0]
    for k in range(len(ball.instances)):          # one draw per ShaderBall instance instead of one instanced draw
        draws.append(bbo.DrawData(ball.vertices, ball.indices, ball.instances[k:k + 1].copy(), ball.material))
    draws += base.draws[1:]
    sc = bbo.Scene(frame, base.view, draws, base.width, base.height)
    img, ref, st = check(sc)
    assert st["n_shaded"] > 10000 and np.isfinite(ref[..., :3]).all()
    # the same frame through the deferred path
    dref, _, dprim, ddepth, _ = bbo.render_deferred(sc, want_gbuffer=False)
    r = Renderer(sc.width, sc.height)
    r.set_option("render_pass", 1)
    r.render_scene(sc)
    assert np.array_equal(r.read_framebuffer().view(np.uint32), dref.view(np.uint32))
    r.close()
