"""Known-answer tests of the oracle's restatement of forward_brdf.frag / brdf.glsl / the fixed-function rules.
The reference holds no tests or golden images, so these closed-form vectors -- the set
SURVEY.md section 8(c) lists -- are what pins rows A2-A6 ("parity unpinned" beyond them: driver latitude)."""
import math

import numpy as np
import pytest

from oracle import bbo, scenes


def const_map(value):
    a = np.zeros((4, 4, 4), np.uint8)
    a[...] = value
    return a


def material(albedo=255, metallic=0, roughness=255, ao=255, normal=(127, 127, 255)):
    return bbo.MaterialData({"albedo": const_map((albedo,) * 3 + (255,)), "metallic": const_map((metallic,) * 3 + (255,)),
                             "roughness": const_map((roughness,) * 3 + (255,)), "ao": const_map((ao,) * 3 + (255,)),
                             "normal": const_map(tuple(normal) + (255,))})


def vary(P=(0, 0, 0), N=(0, 0, 1), uv=(0.5, 0.5), T=(1, 0, 0), B=(0, 1, 0)):
    return np.array(list(uv) + list(P) + list(N) + list(T) + list(B), np.float32)


def view(pos, normal_map=0):
    vu = np.zeros((), bbo.VIEW_DTYPE)
    vu["view_pos"] = pos
    vu["enable_normal_map"] = normal_map
    return vu


def test_kat1_aligned_vectors_roughness_one():
    # N = V = L = H, roughness 1, metallic 0, albedo 1, radiance 1:
    # D = 1/pi, k = 0.5, G = 1, F = 0.04, specular = 0.04/(4 pi), diffuse = 0.96/pi, + ambient 0.03*ao
    fu = scenes.frame_uniforms([scenes.light(2, dir=(0, 0, -1), color=(1, 1, 1), intensity=1.0)])
    out = bbo.shade_fragment(fu, view((0, 0, 5)), material(), vary())
    want = 0.96 / math.pi + 0.04 / (4 * math.pi) + 0.03
    assert out[3] == 1.0
    np.testing.assert_allclose(out[:3], want, rtol=3e-7)
    N = np.array([0, 0, 1], np.float32)
    assert abs(bbo.lib().bbo_distribution_ggx(N.ctypes.data, N.ctypes.data, 1.0) - 1 / math.pi) < 1e-7
    assert abs(bbo.lib().bbo_geometry_smith(N.ctypes.data, N.ctypes.data, N.ctypes.data, 1.0) - 1.0) < 1e-7
    F = np.zeros(3, np.float32); F0 = np.full(3, 0.04, np.float32)
    bbo.lib().bbo_fresnel_schlick(N.ctypes.data, N.ctypes.data, F0.ctypes.data, F.ctypes.data)
    assert np.array_equal(F, F0)


def test_kat2_roughness_zero_off_peak_has_no_specular():
    # roughness 0 and N.H < 1 => D = 0 => only the Lambert term (kD*albedo/pi) * NdotL remains
    L = np.array([0.6, 0.0, 0.8])
    fu = scenes.frame_uniforms([scenes.light(2, dir=tuple(-L), color=(1, 1, 1), intensity=1.0)])
    out = bbo.shade_fragment(fu, view((0, 0, 5)), material(roughness=0), vary())
    H = (L + [0, 0, 1]) / np.linalg.norm(L + [0, 0, 1])
    F = 0.04 + 0.96 * (1 - H[2]) ** 5  # H.V with V = (0,0,1)
    want = (1 - F) / math.pi * 0.8 + 0.03
    np.testing.assert_allclose(out[:3], want, rtol=2e-6)


def test_kat2b_roughness_zero_on_peak_is_nan_like_upstream():
    # brdf.glsl hazard: roughness 0 with N.H = 1 exactly is 0/0 (the `default` material has roughness 0)
    N = np.array([0, 0, 1], np.float32)
    assert np.isnan(bbo.lib().bbo_distribution_ggx(N.ctypes.data, N.ctypes.data, 0.0))
    H = np.array([0.6, 0, 0.8], np.float32)
    assert bbo.lib().bbo_distribution_ggx(N.ctypes.data, H.ctypes.data, 0.0) == 0.0


def test_rsqrt_contract_primitive_accuracy():
    """normalize() uses the contract's fixed Newton inversesqrt: result length within 2 ulp of 1 (GLSL allows 2 ULP)"""
    rng = np.random.Generator(np.random.PCG64(8))
    vu = scenes.view_uniforms((0, 0, 0), 0, 0, 16, 16, 0)
    inst = np.zeros((), bbo.INSTANCE_DTYPE); inst["model"] = np.eye(4); inst["inv_model"] = np.eye(4)
    worst = 0.0
    for _ in range(300):
        v = np.zeros((), bbo.VERTEX_DTYPE)
        v["normal"] = rng.standard_normal(3) * 10 ** rng.uniform(-3, 3)
        v["tangent"] = (1, 0, 0)
        _, vary = bbo.vertex_stage(vu, inst, v)
        n = vary[5:8].astype(np.float64)
        worst = max(worst, abs(np.linalg.norm(n) - 1.0))
        want = v["normal"].astype(np.float64) / np.linalg.norm(v["normal"].astype(np.float64))
        np.testing.assert_allclose(n, want, rtol=0, atol=3e-7)
    assert worst < 2.5e-7


def test_kat3_directional_light_ignores_position():
    a = bbo.shade_fragment(scenes.frame_uniforms([scenes.light(2, pos=(0, 0, 0), dir=(-1, -1, -1), color=(1, .5, .25), intensity=3)]),
                           view((1, 2, 5)), material(roughness=128, metallic=255), vary())
    b = bbo.shade_fragment(scenes.frame_uniforms([scenes.light(2, pos=(99, -7, 3), dir=(-1, -1, -1), color=(1, .5, .25), intensity=3)]),
                           view((1, 2, 5)), material(roughness=128, metallic=255), vary())
    assert np.array_equal(a, b)


def test_kat4_point_light_inverse_square():
    m = material()
    outs = []
    for d in (1.0, 2.0, 4.0):
        fu = scenes.frame_uniforms([scenes.light(0, pos=(0, 0, d), color=(1, 1, 1), intensity=1.0)])
        outs.append(bbo.shade_fragment(fu, view((0, 0, 5)), m, vary())[0] - np.float32(0.03))
    np.testing.assert_allclose(outs[0] / outs[1], 4.0, rtol=1e-5)
    np.testing.assert_allclose(outs[1] / outs[2], 4.0, rtol=1e-5)


def test_spot_light_cutoff_compared_as_given():
    # upstream stores radians in the cut-offs and compares them with a cosine (src/scene.cpp:35-36): keep semantics
    inner, outer = 0.9, 0.8
    for theta_dir, want_scale in (((0, 0, -1), 1.0), ((math.sqrt(1 - 0.85 ** 2), 0, -0.85), 0.5), ((0.8, 0, -0.6), 0.0)):
        spot = scenes.light(1, pos=(0, 0, 2), dir=theta_dir, color=(1, 1, 1), intensity=1.0, inner=inner, outer=outer)
        point = scenes.light(0, pos=(0, 0, 2), color=(1, 1, 1), intensity=1.0)
        a = bbo.shade_fragment(scenes.frame_uniforms([spot]), view((0, 0, 5)), material(), vary())[0] - np.float32(0.03)
        b = bbo.shade_fragment(scenes.frame_uniforms([point]), view((0, 0, 5)), material(), vary())[0] - np.float32(0.03)
        np.testing.assert_allclose(a, b * want_scale, rtol=2e-5, atol=1e-7)


def test_unknown_light_type_contributes_nothing():
    l = scenes.light(7, pos=(0, 0, 1), color=(1, 1, 1), intensity=5.0)
    out = bbo.shade_fragment(scenes.frame_uniforms([l]), view((0, 0, 5)), material(), vary())
    np.testing.assert_allclose(out[:3], 0.03, rtol=1e-6)


def test_normal_map_uses_tbn_and_default_normal_is_flat():
    fu = scenes.frame_uniforms([scenes.light(2, dir=(0, 0, -1), color=(1, 1, 1), intensity=1.0)])
    flat = bbo.shade_fragment(fu, view((0, 0, 5), 0), material(), vary())
    # (127,127,255) decodes to (-1/255, -1/255, 1): nearly +N, so results agree to ~1e-4
    mapped = bbo.shade_fragment(fu, view((0, 0, 5), 1), material(), vary())
    np.testing.assert_allclose(mapped, flat, rtol=2e-4)
    # a tangent-space normal pointing along +T turns the shading normal to T
    tilted = bbo.shade_fragment(fu, view((0, 0, 5), 1), material(normal=(255, 127, 127)), vary())
    assert tilted[0] < flat[0] * 0.2


def test_sampler_texel_centres_wrap_and_bilinear():
    rng = np.random.Generator(np.random.PCG64(5))
    img = rng.integers(0, 256, (8, 16, 4), dtype=np.uint8)
    for (x, y) in ((0, 0), (15, 7), (3, 5)):
        got = bbo.sample(img, 0, (x + 0.5) / 16, (y + 0.5) / 8)
        assert np.array_equal(got, img[y, x].astype(np.float32) * np.float32(1 / 255))
    # REPEAT: u and u+1, u-3 agree at texel centres
    a = bbo.sample(img, 0, (3 + 0.5) / 16, (5 + 0.5) / 8)
    assert np.array_equal(a, bbo.sample(img, 0, (3 + 0.5) / 16 + 1, (5 + 0.5) / 8 - 3))
    # halfway between texel (15, y) and its wrap-around neighbour (0, y)
    got = bbo.sample(img, 0, 0.0, (2 + 0.5) / 8)
    want = (img[2, 15].astype(np.float32) + img[2, 0].astype(np.float32)) * 0.5 / 255
    np.testing.assert_allclose(got, want, rtol=1e-6)
    # non-power-of-two
    img2 = rng.integers(0, 256, (5, 7, 4), dtype=np.uint8)
    assert np.array_equal(bbo.sample(img2, 0, (6 + 0.5) / 7, (4 + 0.5) / 5), img2[4, 6].astype(np.float32) * np.float32(1 / 255))


def test_kat5_triangle_scene_64():
    sc = scenes.triangle_scene(64, 64)
    rgba, prim, depth, st = bbo.render(sc)
    assert st["n_raster_tris"] == 1 and st["n_shaded"] > 0            # front-facing under CLOCKWISE + BACK cull
    ys, xs = np.nonzero(prim != bbo.NO_PRIM)
    top = ys.min()
    assert set(xs[ys == top]) <= {31, 32}                              # apex on the top-centre pixel columns
    assert (np.diff([np.sum(ys == y) for y in range(top, ys.max() + 1)]) >= 0).all()  # widens downwards (Y flipped by P)
    # every covered pixel: default material, directional light from (-1,-1,0) on a (0,0,-1) normal => N.L = 0 => ambient only
    cov = prim != bbo.NO_PRIM
    assert np.array_equal(rgba[cov][:, 3], np.ones(cov.sum(), np.float32))
    np.testing.assert_allclose(rgba[cov][:, :3], 0.03, rtol=1e-6)
    assert (rgba[~cov] == 0).all() and (depth[~cov] == 0).all()
    # reverse-Z: z_view = 5 => ndc z = n/(f-n) * (f - z)/z
    np.testing.assert_allclose(depth[cov], 0.1 / 999.9 * (1000 - 5) / 5, rtol=1e-5)
    # flipped winding is culled
    sc.draws[0].vertices[[1, 2]] = sc.draws[0].vertices[[2, 1]]
    _, prim2, _, st2 = bbo.render(sc)
    assert st2["n_raster_tris"] == 0 and (prim2 == bbo.NO_PRIM).all()


def quad_scene(w, h, z_a, z_b, order):
    """Two overlapping screen-filling quads at view depths z_a / z_b given as two draws in `order`."""
    def quad(z):
        v = np.zeros(4, bbo.VERTEX_DTYPE)
        s = z * 2
        v["pos"] = [(-s, s, z), (s, s, z), (s, -s, z), (-s, -s, z)]
        v["normal"] = (0, 0, -1); v["tangent"] = (1, 0, 0)
        return v
    idx = np.array([0, 1, 2, 2, 3, 0], np.uint32)
    inst = np.zeros(1, bbo.INSTANCE_DTYPE); inst[0]["model"] = np.eye(4); inst[0]["inv_model"] = np.eye(4)
    mats = {"a": bbo.MaterialData({"albedo": const_map((255, 0, 0, 255))}), "b": bbo.MaterialData({"albedo": const_map((0, 255, 0, 255))})}
    z = {"a": z_a, "b": z_b}
    draws = [bbo.DrawData(quad(z[k]), idx, inst, mats[k]) for k in order]
    return bbo.Scene(scenes.frame_uniforms([]), scenes.view_uniforms((0, 0, 0), 0, 0, w, h, 0), draws, w, h)


def test_depth_greater_or_equal_and_api_order_ties():
    # nearer quad wins regardless of draw order (reverse-Z, GREATER_OR_EQUAL)
    for order in ("ab", "ba"):
        rgba, prim, _, _ = bbo.render(quad_scene(16, 16, 2.0, 4.0, order))
        assert (rgba[..., 0] > 0).all() and (rgba[..., 1] == 0).all()
    # exact tie: the later primitive wins (>=)
    rgba, _, _, _ = bbo.render(quad_scene(16, 16, 3.0, 3.0, "ab"))
    assert (rgba[..., 1] > 0).all() and (rgba[..., 0] == 0).all()
    rgba, _, _, _ = bbo.render(quad_scene(16, 16, 3.0, 3.0, "ba"))
    assert (rgba[..., 0] > 0).all() and (rgba[..., 1] == 0).all()


def test_top_left_rule_shared_edges_cover_once():
    # a fan of triangles around a vertex on a pixel centre: every pixel inside the quad is hit exactly once
    sc = quad_scene(32, 32, 3.0, 3.0, "a")
    _, prim, _, st = bbo.render(sc)
    assert st["n_fragments"] == 32 * 32 and st["n_shaded"] == 32 * 32 and (prim != bbo.NO_PRIM).all()


def test_near_plane_clipping_keeps_attributes_continuous():
    # the ShaderBall scene's ground plane spans behind the camera: it must be clipped, not dropped, and the
    # fan of sub-triangles must interpolate like the unclipped plane (world position is linear => check y = -10)
    from bibim_renderer_amd import configs
    cfg = configs.C2.scaled(96, 54, 16)
    sc = scenes.shaderball_scene(cfg, bbo.MaterialData())
    sc.draws = sc.draws[1:]  # plane only
    rgba, prim, depth, st = bbo.render(sc)
    assert st["n_clipped_prims"] == 2 and st["n_shaded"] == 96 * 18  # far edge (z = 50, y = -10) sits 11.3 deg below the horizon
    cov = prim != bbo.NO_PRIM
    assert cov[-1].all() and not cov[0].any()           # ground fills the bottom rows, sky the top
    assert (np.diff(depth[cov[:, 48], 48]) > 0).all()   # depth (reverse-Z) increases towards the viewer = down the screen


def test_forward_shading_equals_visibility_then_shade():
    from bibim_renderer_amd import configs, textures
    mat = bbo.MaterialData(textures.make_material(32))
    sc = scenes.shaderball_scene(configs.C3.scaled(128, 72, 32), mat)
    a, pa, da, _ = bbo.render(sc)
    b, pb, db, _ = bbo.render(sc, flags=bbo.FLAG_FORWARD_SHADE)
    assert np.array_equal(pa, pb) and np.array_equal(da, db)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_row_ranges_compose_to_the_full_frame():
    from bibim_renderer_amd import configs, textures
    mat = bbo.MaterialData(textures.make_material(32))
    sc = scenes.shaderball_scene(configs.C2.scaled(100, 75, 32), mat)
    full, pf, _, st = bbo.render(sc)
    acc = np.zeros_like(full); n = 0
    for y0, y1 in ((0, 13), (13, 14), (14, 60), (60, 75)):
        part, _, _, s = bbo.render(sc, y0, y1)
        acc[y0:y1] = part[y0:y1]; n += s["n_shaded"]
    assert np.array_equal(acc.view(np.uint32), full.view(np.uint32)) and n == st["n_shaded"]


def test_empty_and_degenerate_inputs():
    fu = scenes.frame_uniforms([]); vu = scenes.view_uniforms((0, 0, 0), 0, 0, 8, 8, 0)
    rgba, prim, _, st = bbo.render(bbo.Scene(fu, vu, [], 8, 8))
    assert (rgba == 0).all() and (prim == bbo.NO_PRIM).all() and st["n_prims"] == 0
    v = np.zeros(3, bbo.VERTEX_DTYPE); v["pos"] = [(0, 0, 5), (0, 0, 5), (1, 1, 5)]  # zero area
    inst = np.zeros(1, bbo.INSTANCE_DTYPE); inst[0]["model"] = np.eye(4); inst[0]["inv_model"] = np.eye(4)
    rgba, prim, _, st = bbo.render(bbo.Scene(fu, vu, [bbo.DrawData(v, None, inst, bbo.MaterialData())], 8, 8))
    assert st["n_raster_tris"] == 0 and (prim == bbo.NO_PRIM).all()
    v["pos"] = [(0, 1, -5), (1, -1, -5), (-1, -1, -5)]  # entirely behind the camera
    _, prim, _, st = bbo.render(bbo.Scene(fu, vu, [bbo.DrawData(v, None, inst, bbo.MaterialData())], 8, 8))
    assert (prim == bbo.NO_PRIM).all()


def test_tone_map_formula():
    x = np.array([[0.0, 0.5, 2.0, 0.3]], np.float32)
    on = bbo.tone_map(x, 1, 1.5)
    np.testing.assert_allclose(on[0, :3], 1 - np.exp(-x[0, :3] * 1.5), rtol=1e-6)
    off = bbo.tone_map(x, 0, 1.5)
    assert np.array_equal(off[0, :3], x[0, :3]) and on[0, 3] == 1 and off[0, 3] == 1


def test_band_parallel_render_is_the_whole_frame_render():
    """bbo.render_bands (bands of rows over the host's cores, what the full-size tests and bench.py use) writes the same bits
    as ONE bbo_render call, for band heights that do and do not divide the frame, in both forms of the light loop"""
    from bibim_renderer_amd import configs, textures
    sc = scenes.shaderball_scene(configs.C3.scaled(200, 117, 64), bbo.MaterialData(textures.make_material(64)))
    for flags in (0, bbo.FLAG_LITERAL):
        whole, _, _, st = bbo.render(sc, flags=flags)
        for rows in (32, 7, 117, 500):
            banded, n = bbo.render_bands(sc, flags=flags, rows=rows, threads=3)
            assert n == st["n_shaded"] and np.array_equal(banded.view(np.uint32), whole.view(np.uint32)), (flags, rows)


def test_output_uv_flag_gives_the_winning_fragments_vuv():
    """BBO_FLAG_OUTPUT_UV (measurement aid of tools/texel_lines.py): same coverage as the colour render, alpha 1 on
    geometry, and the interpolated vUV of the winner -- on the TriangleScene a point inside the triangle gets the
    barycentric blend of the three vertex UVs, which stays inside their bounding box"""
    sc = scenes.triangle_scene(64, 64)
    colour, prim, _, _ = bbo.render(sc)
    uv, _, _, _ = bbo.render(sc, flags=bbo.FLAG_OUTPUT_UV)
    cov = prim != bbo.NO_PRIM
    assert np.array_equal(uv[..., 3] == 1.0, cov) and (uv[~cov] == 0).all() and cov.any()
    v = sc.draws[0].vertices["uv"]
    lo, hi = v.min(axis=0) - 1e-6, v.max(axis=0) + 1e-6
    assert (uv[cov][:, 0] >= lo[0]).all() and (uv[cov][:, 0] <= hi[0]).all()
    assert (uv[cov][:, 1] >= lo[1]).all() and (uv[cov][:, 1] <= hi[1]).all() and (uv[cov][:, 2] == 0).all()
    assert len(np.unique(uv[cov][:, :2], axis=0)) > cov.sum() // 2     # (it varies over the triangle)
