"""Mint the committed fixtures under tests/golden/ (authoring container only: reads /root/reference).

  shaderball_vertices.npz  ShaderBall.fbx -> bb::Vertex[29328] (see tools/fbx_geometry.py)
  math_golden.json         outputs of the REFERENCE's vector_math.cpp / camera.cpp (oracle/_ref)
  default_texels.json      resources/pbr/default/*.png decoded by the reference's stb_image 2.25
  gizmo.npz                gizmo.obj/.mtl expanded to bb::GizmoVertex[] + indices
  oracle_frames.npz, n_shaded.json, presented.npz   outputs of the oracle itself, frozen

Fixtures are data (inputs / expected outputs); no reference source text is stored.
"""
from __future__ import annotations

import ctypes as C
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = os.environ.get("BB_REFERENCE", "/root/reference")
GOLD = os.path.join(ROOT, "tests", "golden")


def shaderball():
    from tools.fbx_geometry import load_vertices
    v, info = load_vertices(os.path.join(REF, "resources", "ShaderBall.fbx"))
    info["sha256_f32le"] = hashlib.sha256(np.ascontiguousarray(v).tobytes()).hexdigest()
    np.savez_compressed(os.path.join(ROOT, "bibim_renderer_amd", "data", "shaderball_vertices.npz"), vertices=v)  # package data
    json.dump(info, open(os.path.join(GOLD, "shaderball_vertices.json"), "w"), indent=1)
    print("shaderball", info)


def math_golden():
    ref = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libbb_ref.so"))
    f = C.c_float
    P = C.c_void_p

    def m():
        return np.zeros((4, 4), np.float32)

    def p(a):
        return a.ctypes.data_as(P)

    ref.ref_mat4_translate.argtypes = [f, f, f, P]; ref.ref_mat4_scale.argtypes = [f, f, f, P]
    for n in "xyz":
        getattr(ref, f"ref_mat4_rotate_{n}").argtypes = [f, P]
    ref.ref_mat4_perspective.argtypes = [f, f, f, f, P]
    ref.ref_mat4_mul.argtypes = [P, P, P]; ref.ref_mat4_inverse.argtypes = [P, P]
    ref.ref_mat4_look_at.argtypes = [P, P, P, P]
    ref.ref_camera_look.argtypes = [P, f, f, P]; ref.ref_camera_view.argtypes = [P, f, f, P]

    def bits(a):
        return [int(x) for x in np.ascontiguousarray(a, np.float32).view(np.uint32).ravel()]

    out = {"note": "uint32 bit patterns of float32 results produced by the reference's own "
                   "vector_math.cpp/camera.cpp (g++ -O2 -ffp-contract=off); column-major M[col][row]",
           "cases": []}

    def add(kind, args, result):
        out["cases"].append({"op": kind, "args": args, "bits": bits(result)})

    for fov, aspect, n_, f_ in [(60, 16 / 9, 0.1, 1000), (60, 1920 / 1080, 0.1, 1000), (60, 3840 / 2160, 0.1, 1000),
                                (60, 1.0, 0.1, 1000), (60, 1280 / 720, 0.1, 1000), (45, 4 / 3, 0.5, 50),
                                (90, 2.0, 1.0, 10), (30, 1.0, 0.01, 5000), (75.5, 1.25, 0.2, 300)]:
        o = m(); ref.ref_mat4_perspective(fov, aspect, n_, f_, p(o))
        add("perspective", [float(np.float32(fov)), float(np.float32(aspect)), float(np.float32(n_)), float(np.float32(f_))], o)
    prng = np.random.Generator(np.random.PCG64(77))
    for _ in range(48):  # random fovs: tan(double) vs tanf differ by 1 ulp on some of these
        fov, aspect = float(np.float32(prng.uniform(5, 170))), float(np.float32(prng.uniform(0.2, 4)))
        o = m(); ref.ref_mat4_perspective(fov, aspect, 0.1, 1000.0, p(o))
        add("perspective", [fov, aspect, float(np.float32(0.1)), 1000.0], o)
    for d in (-90, 90, 30, -15, -20, 45, 180, 360, 0.5, 123.456):
        for ax in "xyz":
            o = m(); getattr(ref, f"ref_mat4_rotate_{ax}")(d, p(o))
            add(f"rotate_{ax}", [float(np.float32(d))], o)
    # instance chains: translate * rotateY(-90) * rotateX(-90) * scale(0.01), and their inverses
    for (tx, ty, tz) in [(0, -1, 2), (2, -1, 2), (-3, -1, 2), (1, -1, 8), (7, -1, 16), (0, -10, 0)]:
        t = m(); ry = m(); rx = m(); s = m(); a = m(); b = m(); c = m(); inv = m()
        ref.ref_mat4_translate(tx, ty, tz, p(t)); ref.ref_mat4_rotate_y(-90, p(ry)); ref.ref_mat4_rotate_x(-90, p(rx))
        ref.ref_mat4_scale(0.01, 0.01, 0.01, p(s))
        ref.ref_mat4_mul(p(t), p(ry), p(a)); ref.ref_mat4_mul(p(a), p(rx), p(b)); ref.ref_mat4_mul(p(b), p(s), p(c))
        ref.ref_mat4_inverse(p(c), p(inv))
        add("instance_chain", [float(tx), float(ty), float(tz)], c)
        add("instance_chain_inverse", [float(tx), float(ty), float(tz)], inv)
    t = m(); s = m(); c = m(); inv = m()
    ref.ref_mat4_translate(0, -10, 0, p(t)); ref.ref_mat4_scale(100, 100, 100, p(s)); ref.ref_mat4_mul(p(t), p(s), p(c))
    ref.ref_mat4_inverse(p(c), p(inv))
    add("plane_model", [], c); add("plane_model_inverse", [], inv)
    rng = np.random.Generator(np.random.PCG64(1234))
    for _ in range(8):
        a = rng.standard_normal((4, 4)).astype(np.float32); b = rng.standard_normal((4, 4)).astype(np.float32)
        o = m(); ref.ref_mat4_mul(p(a), p(b), p(o)); add("mul", [bits(a), bits(b)], o)
        o = m(); ref.ref_mat4_inverse(p(a), p(o)); add("inverse", [bits(a)], o)
    for pos, yaw, pitch in [((0, 0, 0), 0, 0), ((0, 2, -2), 0, -15), ((0, 4, -6), 0, -20), ((1, 2, 3), 30, 10),
                            ((-5, 0.5, 2), -120, 45), ((0, 0, 0), 90, -89)]:
        pa = np.asarray(pos, np.float32)
        o3 = np.zeros(3, np.float32); ref.ref_camera_look(p(pa), yaw, pitch, p(o3))
        add("camera_look", [float(yaw), float(pitch)], o3)
        o = m(); ref.ref_camera_view(p(pa), yaw, pitch, p(o))
        add("camera_view", [[float(x) for x in pos], float(yaw), float(pitch)], o)
    for eye, tgt, up in [((0, 0, -5), (0, 0, 0), (0, 1, 0)), ((3, 4, 5), (-1, 0.5, 2), (0, 1, 0)), ((0, 10, 0), (1, 0, 1), (0, 0, 1))]:
        o = m(); ref.ref_mat4_look_at(p(np.asarray(eye, np.float32)), p(np.asarray(tgt, np.float32)), p(np.asarray(up, np.float32)), p(o))
        add("look_at", [list(map(float, eye)), list(map(float, tgt)), list(map(float, up))], o)
    json.dump(out, open(os.path.join(GOLD, "math_golden.json"), "w"))
    print("math_golden", len(out["cases"]), "cases")


def default_texels():
    stb = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libstb_ref.so"))
    stb.stbi_load.restype = C.POINTER(C.c_ubyte)
    stb.stbi_load.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int]
    out = {"note": "resources/pbr/default/<map>.png decoded with the reference's stb_image 2.25, STBI_rgb_alpha "
                   "(src/resource.cpp:159-160); every file is a uniform image", "maps": {}}
    d = os.path.join(REF, "resources", "pbr", "default")
    for name in ("albedo", "metallic", "roughness", "ao", "normal", "height"):
        path = os.path.join(d, name + ".png")
        if not os.path.exists(path):
            out["maps"][name] = None
            continue
        w, h, ch = C.c_int(), C.c_int(), C.c_int()
        px = stb.stbi_load(path.encode(), C.byref(w), C.byref(h), C.byref(ch), 4)
        a = np.ctypeslib.as_array(px, shape=(h.value, w.value, 4)).copy()
        uniform = bool((a == a[0, 0]).all())
        out["maps"][name] = {"w": w.value, "h": h.value, "file_channels": ch.value, "uniform": uniform,
                             "texel": [int(x) for x in a[0, 0]]}
    json.dump(out, open(os.path.join(GOLD, "default_texels.json"), "w"), indent=1)
    print("default_texels", out["maps"])


def gizmo():
    from tools.obj_loader import load_gizmo
    v, idx, info = load_gizmo(os.path.join(REF, "resources", "gizmo.obj"))
    np.savez_compressed(os.path.join(GOLD, "gizmo.npz"), vertices=v, indices=idx)
    json.dump(info, open(os.path.join(GOLD, "gizmo.json"), "w"), indent=1)
    print("gizmo", info)


def golden_frames():
    """Oracle outputs frozen as regression fixtures (so a change of the contract is a visible diff) and as
    the CPU-side expected values the GPU tests compare against at sizes with no oracle run."""
    from bibim_renderer_amd import configs, textures
    from oracle import bbo, scenes

    def pack(rgba, prim, depth):
        return {"rgba_bits": rgba.view(np.uint32), "prim": prim, "depth_bits": depth.view(np.uint32)}

    out = {}
    rgba, prim, depth, st = bbo.render(scenes.triangle_scene(64, 64))
    out.update({f"triangle64_{k}": v for k, v in pack(rgba, prim, depth).items()})
    mat = bbo.MaterialData(textures.make_material(64))
    rgba, prim, depth, st2 = bbo.render(scenes.shaderball_scene(configs.C2.scaled(160, 90, 64), mat))
    out.update({f"c2_160x90_{k}": v for k, v in pack(rgba, prim, depth).items()})
    # the same frames with the light loop evaluated statement by statement as the GLSL is written (BBO_FLAG_LITERAL);
    # the default above is the shipped evaluation order, which the GPU reproduces bit for bit
    lit, _, _, _ = bbo.render(scenes.triangle_scene(64, 64), flags=bbo.FLAG_LITERAL)
    out["triangle64_literal_rgba_bits"] = lit.view(np.uint32)
    lit, _, _, _ = bbo.render(scenes.shaderball_scene(configs.C2.scaled(160, 90, 64), mat), flags=bbo.FLAG_LITERAL)
    out["c2_160x90_literal_rgba_bits"] = lit.view(np.uint32)
    g = np.load(os.path.join(GOLD, "gizmo.npz"))
    v = np.zeros(len(g["vertices"]), bbo.GIZMO_VERTEX_DTYPE)
    v["pos"], v["color"], v["normal"] = g["vertices"][:, 0:3], g["vertices"][:, 3:6], g["vertices"][:, 6:9]
    vu = scenes.view_uniforms((0, 0, 0), 0.0, 0.0, 256, 256, 0)  # C1: identity camera rotation
    rgba, prim, depth, st3 = bbo.render_gizmo(vu, v, g["indices"], 256, 256)
    out.update({f"gizmo256_{k}": v_ for k, v_ in pack(rgba, prim, depth).items()})
    np.savez_compressed(os.path.join(GOLD, "oracle_frames.npz"), **out)
    json.dump({"triangle64": st, "c2_160x90": st2, "gizmo256": st3}, open(os.path.join(GOLD, "oracle_frames.json"), "w"), indent=1)
    print("golden_frames", st, st2, st3)


def n_shaded():
    """N_shaded per BASELINE config (counted by the CPU oracle, stored with the fixtures)."""
    from bibim_renderer_amd import configs, textures
    from oracle import bbo, scenes
    mat = bbo.MaterialData(textures.make_material(64))  # coverage does not depend on the texels
    out = {}
    for name in ("c2", "c3", "c5"):
        cfg = configs.CONFIGS[name]
        _, _, _, st = bbo.render(scenes.shaderball_scene(cfg, mat), flags=0, want_prim=False, want_depth=False)
        out[name] = st
        print(name, st)
    json.dump(out, open(os.path.join(GOLD, "n_shaded.json"), "w"), indent=1)


def reference_pngs():
    """Every PNG the reference ships, decoded with the reference's own stb_image 2.25 (oracle/_ref/libstb_ref.so,
    STBI_rgb_alpha): size + sha256 of the RGBA8 bytes.  Pins the in-repo PNG decoder (bba_load_png)."""
    import glob
    stb = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libstb_ref.so"))
    stb.stbi_load.restype = C.POINTER(C.c_ubyte)
    stb.stbi_load.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int]
    stb.stbi_image_free.argtypes = [C.c_void_p]
    out = {}
    for path in sorted(glob.glob(os.path.join(REF, "resources", "**", "*.png"), recursive=True)):
        w, h, ch = C.c_int(), C.c_int(), C.c_int()
        px = stb.stbi_load(path.encode(), C.byref(w), C.byref(h), C.byref(ch), 4)
        if not px:
            continue
        a = np.ctypeslib.as_array(px, shape=(h.value, w.value, 4))
        out[os.path.relpath(path, os.path.join(REF, "resources"))] = {
            "w": w.value, "h": h.value, "file_channels": ch.value, "sha256_rgba8": hashlib.sha256(a.tobytes()).hexdigest()}
        stb.stbi_image_free(px)
    json.dump(out, open(os.path.join(GOLD, "reference_png_sha256.json"), "w"), indent=1)
    print("reference_pngs", len(out), "files")


def uv_sphere():
    """The light-marker sphere (generateUVSphereMesh(0.1, 16, 16), src/main.cpp:953-957): every position computed by the
    REFERENCE's own sphericalToCartesian (oracle/_ref/libbb_ref.so) from the angles of src/render.cpp:1805-1812, plus the
    index pattern of :1825-1838 restated here.  Pins bbo_uv_sphere and the library's marker mesh."""
    ref = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libbb_ref.so"))
    ref.ref_spherical_to_cartesian.argtypes = [C.c_float, C.c_float, C.c_float, C.POINTER(C.c_float)]
    f32 = np.float32
    pi32 = f32(3.141592)
    half_pi, two_pi = pi32 * f32(0.5), pi32 * f32(2.0)
    H = V = 16
    pos = np.zeros(((H + 1) * (V + 1), 3), np.float32)
    k = 0
    for v in range(V + 1):
        theta = f32(-half_pi + pi32 * (f32(v) / f32(V)))
        for h in range(H + 1):
            phi = f32(two_pi * (f32(h) / f32(H)))
            out = (C.c_float * 3)()
            ref.ref_spherical_to_cartesian(f32(0.1), theta, phi, out)
            pos[k] = out[:]
            k += 1
    idx = []
    for v in range(V):
        for h in range(H):
            base = (H + 1) * v + h
            if v < V - 1:
                idx += [base, base + H + 1, base + H + 2]
            if v > 0:
                idx += [base + H + 2, base + 1, base]
    idx = np.array(idx, np.uint32)
    np.savez_compressed(os.path.join(GOLD, "uv_sphere.npz"), pos_bits=pos.view(np.uint32), indices=idx)
    print("uv_sphere", pos.shape, idx.shape, hashlib.sha256(pos.tobytes()).hexdigest()[:16])


def overlays():
    """Overlay subpass of the oracle frozen: markers + gizmo over the presented golden C2 160x90 frame."""
    from bibim_renderer_amd import configs, textures
    from oracle import bbo, scenes
    mat = bbo.MaterialData(textures.make_material(64))
    sc = scenes.shaderball_scene(configs.C2.scaled(160, 90, 64), mat)
    sc.frame = scenes.frame_uniforms(scenes.reference_default_lights(), 1, 1.7)
    hdr, _, depth, _ = bbo.render(sc)
    base = bbo.present(hdr, 1, 1.7)
    g = np.load(os.path.join(GOLD, "gizmo.npz"))
    gv = np.zeros(len(g["vertices"]), bbo.GIZMO_VERTEX_DTYPE)
    gv["pos"], gv["color"], gv["normal"] = g["vertices"][:, 0:3], g["vertices"][:, 3:6], g["vertices"][:, 6:9]
    img, st = bbo.overlay(sc.frame, sc.view, depth, base, gv, g["indices"], 48)
    np.savez_compressed(os.path.join(GOLD, "overlays.npz"), c2_160x90_base=base, c2_160x90_overlaid=img)
    json.dump({"stats": st, "gizmo_extent": 48, "sha256": hashlib.sha256(img.tobytes()).hexdigest()},
              open(os.path.join(GOLD, "overlays.json"), "w"), indent=1)
    print("overlays", st)


def present():
    """Presentation contract frozen: the 255 sRGB thresholds (bit patterns) and the presented bytes of the golden
    C2 160x90 frame with and without tone mapping (sha256 + the image itself, 57 KB each)."""
    from oracle import bbo
    thr = bbo.srgb_thresholds()
    frames = np.load(os.path.join(GOLD, "oracle_frames.npz"))
    hdr = frames["c2_160x90_rgba_bits"].view(np.float32)
    out = {"srgb_thresholds_bits": thr.view(np.uint32)}
    info = {"thresholds_sha256": hashlib.sha256(thr.tobytes()).hexdigest()}
    for tag, enable, exposure, hdr16 in (("plain", 0, 1.0, 1), ("tonemapped", 1, 1.7, 1), ("tonemapped_fp32", 1, 1.7, 0)):
        img = bbo.present(hdr, enable, exposure, hdr16)
        out[f"c2_160x90_{tag}"] = img
        info[tag] = {"enable": enable, "exposure": exposure, "hdr16": hdr16, "sha256": hashlib.sha256(img.tobytes()).hexdigest()}
    np.savez_compressed(os.path.join(GOLD, "presented.npz"), **out)
    json.dump(info, open(os.path.join(GOLD, "presented.json"), "w"), indent=1)
    print("present", info)


def deferred():
    """Deferred path of the oracle frozen: colour bits, G-buffer (as binary16) and coverage of the golden C2 160x90 scene."""
    from bibim_renderer_amd import configs, textures
    from oracle import bbo, scenes
    mat = bbo.MaterialData(textures.make_material(64))
    rgba, gbuf, prim, depth, st = bbo.render_deferred(scenes.shaderball_scene(configs.C2.scaled(160, 90, 64), mat))
    g16 = gbuf.astype(np.float16)
    assert np.array_equal(g16.astype(np.float32), gbuf)
    lit = bbo.render_deferred(scenes.shaderball_scene(configs.C2.scaled(160, 90, 64), mat), flags=bbo.FLAG_LITERAL)[0]
    np.savez_compressed(os.path.join(GOLD, "deferred.npz"), c2_160x90_rgba_bits=rgba.view(np.uint32), c2_160x90_gbuffer_f16=g16,
                        c2_160x90_prim=prim, c2_160x90_depth_bits=depth.view(np.uint32),
                        c2_160x90_literal_rgba_bits=lit.view(np.uint32))
    info = {"c2_160x90": st, "rgba_sha256": hashlib.sha256(rgba.tobytes()).hexdigest(),
            "gbuffer_sha256": hashlib.sha256(g16.tobytes()).hexdigest()}
    json.dump(info, open(os.path.join(GOLD, "deferred.json"), "w"), indent=1)
    print("deferred", info)


def contract_manifest():
    """tests/golden/CONTRACT.json: which revision of the arithmetic contract (oracle/bb_oracle.h, BBO_CONTRACT_REVISION)
    minted the frozen frames, and their hashes -- a re-mint shows up as a diff of this file next to a new revision."""
    from oracle import bbo
    frozen = ["oracle_frames.npz", "deferred.npz", "presented.npz", "overlays.npz"]
    path = os.path.join(GOLD, "CONTRACT.json")
    old = json.load(open(path)) if os.path.exists(path) else {"history": []}
    rev = bbo.contract_revision()
    files = {f: hashlib.sha256(open(os.path.join(GOLD, f), "rb").read()).hexdigest() for f in frozen}
    if old.get("files") not in (None, files) and old.get("contract_revision") == rev:
        # frozen frames that changed WITHOUT a new revision: exactly what this manifest exists to make impossible
        raise SystemExit(f"the frozen frames differ from the ones contract revision {rev} minted: bump BBO_CONTRACT_REVISION "
                         "(oracle/bb_oracle.h, with a line in its history) before re-minting, or restore the fixtures")
    if old.get("contract_revision") != rev:
        old["history"] = old.get("history", []) + [{"contract_revision": rev, "files": files}]
    old.update({"contract_revision": rev, "files": files,
                "note": "frames frozen from the oracle's CONTRACT (default) form; keys *_literal_rgba_bits inside them are the "
                        "statement-by-statement form, which has no revision.  tests/test_oracle_golden_frames.py refuses a "
                        "fixture whose hash or revision differs from this file."})
    json.dump(old, open(path, "w"), indent=1)
    print("contract manifest", rev)


if __name__ == "__main__":
    os.makedirs(GOLD, exist_ok=True)
    which = sys.argv[1:] or ["shaderball", "math_golden", "default_texels", "gizmo", "golden_frames", "n_shaded", "present", "deferred", "reference_pngs", "uv_sphere", "overlays"]
    if sys.argv[1:] == ["contract_manifest"]:
        contract_manifest()
        sys.exit(0)
    for w in which:
        globals()[w]()
    contract_manifest()
