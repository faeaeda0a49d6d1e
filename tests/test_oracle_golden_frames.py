"""The oracle reproduces its committed golden framebuffers bit-for-bit (tests/golden/oracle_frames.npz, minted by
tools/make_fixtures.py): a change of the arithmetic contract shows up as a fixture diff, and the GPU box checks
the same bits without /root/reference."""
import json
import os

import numpy as np

from conftest import GOLDEN
from bibim_renderer_amd import configs, textures
from oracle import bbo, scenes


def _g():
    return np.load(os.path.join(GOLDEN, "oracle_frames.npz")), json.load(open(os.path.join(GOLDEN, "oracle_frames.json")))


def _check(prefix, rgba, prim, depth, st):
    g, stats = _g()
    assert np.array_equal(prim, g[prefix + "_prim"])
    assert np.array_equal(depth.view(np.uint32), g[prefix + "_depth_bits"])
    assert np.array_equal(rgba.view(np.uint32), g[prefix + "_rgba_bits"])
    assert st == stats[prefix]


def _check_literal(prefix, scene):
    """both forms of the light loop are frozen: the shipped evaluation order (default) and the statement-by-statement
    one (BBO_FLAG_LITERAL); they share coverage and depth and agree within BASELINE's tolerance"""
    g, _ = _g()
    lit = bbo.render(scene, flags=bbo.FLAG_LITERAL)[0]
    assert np.array_equal(lit.view(np.uint32), g[prefix + "_literal_rgba_bits"])
    ref = g[prefix + "_rgba_bits"].view(np.float32)
    with np.errstate(invalid="ignore"):
        ok = np.abs(lit - ref) <= 1e-4 * np.maximum(1.0, np.abs(lit))
    assert np.all(ok | (np.isnan(lit) & np.isnan(ref)))


def test_triangle_scene_golden():
    _check("triangle64", *bbo.render(scenes.triangle_scene(64, 64)))
    _check_literal("triangle64", scenes.triangle_scene(64, 64))


def test_c2_small_golden():
    mat = bbo.MaterialData(textures.make_material(64))
    sc = scenes.shaderball_scene(configs.C2.scaled(160, 90, 64), mat)
    _check("c2_160x90", *bbo.render(sc))
    _check_literal("c2_160x90", sc)


def gizmo_inputs():
    g = np.load(os.path.join(GOLDEN, "gizmo.npz"))
    v = np.zeros(len(g["vertices"]), bbo.GIZMO_VERTEX_DTYPE)
    v["pos"], v["color"], v["normal"] = g["vertices"][:, 0:3], g["vertices"][:, 3:6], g["vertices"][:, 6:9]
    return v, g["indices"]


def test_c1_gizmo_256_golden_and_properties():
    """BASELINE config #1: gizmo.obj, flat Lambert, 256x256, CPU reference rasteriser only."""
    v, idx = gizmo_inputs()
    info = json.load(open(os.path.join(GOLDEN, "gizmo.json")))
    assert info["triangles"] == 594 and info["polygons_by_size"] == {"3": 180, "4": 180, "20": 3}
    vu = scenes.view_uniforms((0, 0, 0), 0.0, 0.0, 256, 256, 0)
    rgba, prim, depth, st = bbo.render_gizmo(vu, v, idx, 256, 256)
    _check("gizmo256", rgba, prim, depth, st)
    cov = prim != bbo.NO_PRIM
    # colours are Kd * diffuse with Kd in {grey, red, green, blue}: every covered pixel is a scaled Kd
    kd = np.array([[.5, .5, .5], [1, 0, 0], [0, 1, 0], [0, 0, 1]], np.float32)
    px = rgba[cov][:, :3]
    scale = px.max(1, keepdims=True)
    lit = scale[:, 0] > 0
    unit = px[lit] / scale[lit]
    d = np.abs(unit[:, None, :] - (kd / kd.max(1, keepdims=True))[None]).max(2).min(1)
    assert (d < 1e-6).all()
    assert (rgba[cov][:, 3] == 1).all() and (rgba[~cov] == 0).all()
    # identity camera: +X (red) arrow points right of centre, +Y (green) arrow up
    ys, xs = np.nonzero(cov & (rgba[..., 0] > 0) & (rgba[..., 1] == 0) & (rgba[..., 2] == 0))
    assert xs.mean() > 140
    ys, xs = np.nonzero(cov & (rgba[..., 1] > 0) & (rgba[..., 0] == 0) & (rgba[..., 2] == 0))
    assert ys.mean() < 116


def test_n_shaded_fixture_matches_oracle_at_c2():
    want = json.load(open(os.path.join(GOLDEN, "n_shaded.json")))
    mat = bbo.MaterialData(textures.make_material(16))
    _, _, _, st = bbo.render(scenes.shaderball_scene(configs.C2, mat), want_prim=False, want_depth=False)
    assert st == want["c2"]


def test_texture_generator_is_deterministic():
    import hashlib
    m = textures.make_material(64)
    h = hashlib.sha256(b"".join(m[k].tobytes() for k in ("albedo", "metallic", "roughness", "ao", "normal"))).hexdigest()
    assert h == open(os.path.join(GOLDEN, "textures64.sha256")).read().strip()
    assert m["roughness"][..., 0].min() >= 39  # never 0: brdf.glsl's 0/0 hazard stays out of the benchmark


def test_frozen_frames_carry_the_contract_revision_that_minted_them():
    """tests/golden/CONTRACT.json (tools/make_fixtures.py contract_manifest): the frozen frames are the ones that file
    lists, minted by the contract revision the oracle is built with -- a re-mint without a new BBO_CONTRACT_REVISION, or a
    new revision without re-minted frames, fails here instead of passing silently."""
    import hashlib
    m = json.load(open(os.path.join(GOLDEN, "CONTRACT.json")))
    assert m["contract_revision"] == bbo.contract_revision()
    assert m["history"][-1]["contract_revision"] == m["contract_revision"] and m["history"][-1]["files"] == m["files"]
    revs = [h["contract_revision"] for h in m["history"]]
    assert all(a < b for a, b in zip(revs, revs[1:])), "one history entry per revision, strictly increasing: a re-mint needs a new revision"
    for name, digest in m["files"].items():
        assert hashlib.sha256(open(os.path.join(GOLDEN, name), "rb").read()).hexdigest() == digest, name


def test_the_threaded_oracle_renders_the_same_frame(maps64):
    """bbo_render_parallel (what bench.py's all-cores CPU baseline times): primitives set up once, bands from a queue -- the frame,
    the shaded-pixel count and the fragment count of bbo_render, bit for bit, in both forms of the light loop, for any number of
    threads and band heights that do not divide the frame"""
    import numpy as np
    from bibim_renderer_amd import configs
    from oracle import bbo, scenes
    sc = scenes.shaderball_scene(configs.C3.scaled(480, 270, 64), bbo.MaterialData(maps64))
    ref, _, _, st = bbo.render(sc)
    for threads, rows in ((1, 8), (3, 7), (8, 32), (5, 300)):
        got, st2 = bbo.render_parallel(sc, threads=threads, rows=rows)
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), (threads, rows)
        assert all(st2[k] == st[k] for k in ("n_prims", "n_raster_tris", "n_clipped_prims", "n_fragments", "n_shaded")), (st, st2)
    lit, _ = bbo.render_bands(sc, flags=bbo.FLAG_LITERAL)
    got, _ = bbo.render_parallel(sc, flags=bbo.FLAG_LITERAL, threads=4)
    assert np.array_equal(got.view(np.uint32), lit.view(np.uint32))
    # an empty frame and the clipped ground plane alone
    sc.draws = sc.draws[1:]
    ref, _, _, st = bbo.render(sc)
    got, st2 = bbo.render_parallel(sc, threads=4, rows=16)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)) and st2["n_clipped_prims"] == st["n_clipped_prims"] > 0
