"""What pins the oracle's ARITHMETIC.

The reference holds no golden frame and cannot run here (Vulkan + glslc), so the oracle's shading arithmetic is pinned by
two things written in this file, independently of oracle/bb_oracle.c:

  * `glsl_f64_*`: the statements of src/shaders/brdf.glsl:2-36 and src/shaders/forward_brdf.frag:27-75 typed again from
    the GLSL, evaluated in binary64 with numpy (real divisions, real pow, real normalize).  The oracle's binary32 LITERAL
    form must agree with it to 1e-5 relative wherever the expression is well conditioned, and within the conditioning
    bound everywhere else (the GGX denominator NdotH^2 (a2 - 1) + 1 cancels near the specular peak of a smooth surface:
    a binary32 evaluation of the shader -- any, the reference's own SPIR-V included -- carries eps32 / denominator there).
  * `np_bilinear`: VK_FILTER_LINEAR + REPEAT addressing on an RGBA8 UNORM image typed from the Vulkan specification's
    texel-coordinate rules (unnormalised = u * size, texel centre at +0.5, weights = frac), in binary64.

and by the relation between the oracle's two forms:

  * the CONTRACT form (the evaluation order the GPU ships: one reciprocal for the specular term, hoisted diffuse factor,
    saturate for the clamped cosines; oracle/bb_oracle.c `light_surface_contract`) against the LITERAL form (statement by
    statement): <= 1e-5 relative on well-conditioned random surfaces, <= 1e-4 * max(1, |ref|) on whole frames
    (BASELINE's tolerance).

The GPU is bit-exact against the contract form (tests/test_gpu_parity.py); this file is what ties the contract form to
the GLSL."""
import numpy as np
import pytest

from bibim_renderer_amd import configs
from oracle import bbo, scenes

N_SAMPLES = 200_000
WELL = 0.1            # conditioning() >= WELL: "away from the hazards"
BOUND_EPS = 2.0 ** -20  # a few eps32: the relative error a binary32 evaluation may carry per unit of 1 / conditioning()
PI = 3.1415926535897932384626433832795  # brdf.glsl:2


# ---------------------------------------------------------------------------------------------------------------------
# binary64 evaluation typed from the GLSL
# ---------------------------------------------------------------------------------------------------------------------
def _dot(a, b):
    return (a * b).sum(-1)


def _normalize(a):
    return a / np.sqrt(_dot(a, a))[..., None]


def glsl_f64_distribution_ggx(N, H, roughness):  # brdf.glsl:5-17
    a = roughness * roughness
    a2 = a * a
    NdotH = np.maximum(_dot(N, H), 0)
    NdotH2 = NdotH * NdotH
    num = a2
    denom = NdotH2 * (a2 - 1) + 1
    denom = PI * denom * denom
    return num / denom


def glsl_f64_geometry_schlick_ggx(NdotV, roughness):  # brdf.glsl:19-25
    r = roughness + 1
    k = (r * r) / 8
    num = NdotV
    denom = NdotV * (1 - k) + k
    return num / denom


def glsl_f64_geometry_smith(N, V, L, k):  # brdf.glsl:27-33
    NdotV = np.maximum(_dot(N, V), 0)
    NdotL = np.maximum(_dot(N, L), 0)
    return glsl_f64_geometry_schlick_ggx(NdotV, k) * glsl_f64_geometry_schlick_ggx(NdotL, k)


def glsl_f64_fresnel_schlick(H, V, F0):  # brdf.glsl:35-37
    return F0 + (1 - F0) * np.power(1 - np.maximum(_dot(H, V), 0), 5)[..., None]


def glsl_f64_light_loop(lights, view_pos, P, normal, albedo, metallic, roughness, ao):
    """forward_brdf.frag:27-75 on n surface points at once (arrays [n, 3] / [n]); lights: list of dicts."""
    Lo = np.zeros_like(P)
    for light in lights:  # :29
        if light["type"] == 0:  # :33-37
            L = light["pos"] - P
            d = np.sqrt(_dot(L, L))
            att = 1 / (d * d)
            L = _normalize(L)
        elif light["type"] == 1:  # :38-46
            L = light["pos"] - P
            d = np.sqrt(_dot(L, L))
            att = 1 / (d * d)
            L = _normalize(L)
            theta = _dot(L, _normalize(-light["dir"]))
            epsilon = light["inner"] - light["outer"]
            att = att * np.clip((theta - light["outer"]) / epsilon, 0, 1)
        elif light["type"] == 2:  # :47-50
            L = np.broadcast_to(-_normalize(light["dir"]), P.shape)
            att = np.ones(len(P))
        else:
            raise ValueError("L and att are undefined in the GLSL for other types")
        V = _normalize(view_pos - P)  # :52
        N = _normalize(normal)
        H = _normalize(L + V)
        D = glsl_f64_distribution_ggx(N, H, roughness)  # :56
        F0 = np.full_like(P, 0.04)
        F0 = F0 * (1 - metallic[:, None]) + albedo * metallic[:, None]  # mix(x, y, a) = x (1 - a) + y a
        F = glsl_f64_fresnel_schlick(H, V, F0)
        G = glsl_f64_geometry_smith(N, V, L, roughness)
        radiance = att[:, None] * light["color"] * light["intensity"]  # :63
        specular = (D[:, None] * F * G[:, None]) / np.maximum(
            4 * np.maximum(_dot(V, N), 0) * np.maximum(_dot(L, N), 0), 0.001)[:, None]  # :65
        kS = F
        kD = 1 - kS
        kD = kD * (1 - metallic[:, None])
        Lo = Lo + (kD * albedo / PI + specular) * radiance * np.maximum(_dot(N, L), 0)[:, None]  # :70
    ambient = 0.03 * albedo * ao[:, None]  # :73
    return ambient + Lo


def conditioning(lights, view_pos, P, normal, roughness):
    """How far every surface point is from the places where the shader's value is ill-conditioned in ITS OWN inputs, so that
    any binary32 evaluation of it (the reference's SPIR-V included) carries eps32 / h of relative error:
      * the GGX denominator NdotH^2 (a2 - 1) + 1, which cancels at the specular peak of a smooth surface (0/0 at
        roughness 0, brdf.glsl:12),
      * a clamped cosine just above its clamp: NdotV, NdotL (a grazing view or light: the cosine's own rounding error is
        eps32 absolute, and below max(4 NdotV NdotL, 0.001) it no longer cancels out of specular, forward_brdf.frag:65),
      * a spot light's cone factor just inside the outer edge ((theta - outerCutOff) cancels, :45).
    h = the smallest of those quantities over the lights."""
    h = np.ones(len(P))
    a2 = roughness ** 4
    N = _normalize(normal)
    V = _normalize(view_pos - P)
    ndv = _dot(N, V)
    h = np.where(ndv > 0, np.minimum(h, ndv), h)
    for light in lights:
        L = np.broadcast_to(-_normalize(light["dir"]), P.shape) if light["type"] == 2 else _normalize(light["pos"] - P)
        H = _normalize(L + V)
        ndh = np.maximum(_dot(N, H), 0)
        h = np.minimum(h, ndh * ndh * (a2 - 1) + 1)
        ndl = _dot(N, L)
        h = np.where(ndl > 0, np.minimum(h, ndl), h)
        if light["type"] == 1:
            x = (_dot(L, _normalize(-light["dir"])) - light["outer"]) / (light["inner"] - light["outer"])
            h = np.where((x > 0) & (x < 1), np.minimum(h, x), h)
    return h


# ---------------------------------------------------------------------------------------------------------------------
# random surfaces and lights (binary32 values, so that both evaluations start from the same real numbers)
# ---------------------------------------------------------------------------------------------------------------------
def random_case(seed, n, n_lights=4, types=(0, 1, 2)):
    rng = np.random.Generator(np.random.PCG64(seed))
    f32 = lambda a: np.asarray(a, np.float32)
    P = f32(rng.uniform(-4, 4, (n, 3)))
    normal = f32(rng.normal(size=(n, 3)))
    normal[np.abs(normal).sum(1) < 0.2] = (0, 1, 0)
    albedo = f32(rng.uniform(0, 1, (n, 3)))
    metallic = f32(rng.uniform(0, 1, n))
    roughness = f32(rng.uniform(0.02, 1, n))
    ao = f32(rng.uniform(0, 1, n))
    lights = []
    for i in range(n_lights):
        d = f32(rng.normal(size=3))
        outer = np.float32(rng.uniform(0.3, 0.8))
        lights.append(dict(type=int(types[i % len(types)]), pos=f32(rng.uniform(-9, 9, 3) + (0, 12, 0)), dir=d,
                           color=f32(rng.uniform(0.1, 1, 3)), intensity=np.float32(rng.uniform(1, 200)),
                           inner=np.float32(outer + rng.uniform(0.05, 0.19)), outer=outer))
    view_pos = f32((0.5, 6.0, 9.0))
    surf = np.concatenate([P, normal, albedo, metallic[:, None], roughness[:, None], ao[:, None]], axis=1)
    frame = scenes.frame_uniforms([scenes.light(l["type"], pos=l["pos"], dir=l["dir"], color=l["color"],
                                                intensity=float(l["intensity"]), inner=float(l["inner"]),
                                                outer=float(l["outer"])) for l in lights])
    view = scenes.view_uniforms(tuple(float(x) for x in view_pos), 0.0, 0.0, 64, 64, 0)
    f64 = lambda a: np.asarray(a, np.float64)
    lights64 = [{k: (f64(v) if isinstance(v, np.ndarray) else (float(v) if k != "type" else v)) for k, v in l.items()}
                for l in lights]
    args64 = (lights64, f64(view_pos), f64(P), f64(normal), f64(albedo), f64(metallic), f64(roughness), f64(ao))
    return frame, view, surf, args64


@pytest.fixture(scope="module")
def case():
    frame, view, surf, args64 = random_case(2024, N_SAMPLES)
    want = glsl_f64_light_loop(*args64)
    literal = bbo.light_surface(frame, view, surf, literal=True)[:, :3].astype(np.float64)
    contract = bbo.light_surface(frame, view, surf, literal=False)[:, :3].astype(np.float64)
    h = conditioning(args64[0], args64[1], args64[2], args64[3], args64[6])
    return want, literal, contract, h


def rel_err(got, want):
    return np.abs(got - want).max(1) / np.maximum(np.abs(want).max(1), 1e-30)


def test_literal_form_against_the_glsl_in_binary64(case):
    want, literal, _, h = case
    assert np.isfinite(want).all() and np.isfinite(literal).all()
    err = rel_err(literal, want)
    well = h >= WELL  # away from the hazards: see conditioning()
    assert well.sum() >= 100_000, int(well.sum())
    assert err[well].max() <= 1e-5, float(err[well].max())
    # everywhere else: within the conditioning bound (the GGX denominator's error is squared into D)
    assert (err <= 1e-5 + BOUND_EPS / h).all(), float((err / (1e-5 + BOUND_EPS / h)).max())


def test_contract_form_against_the_glsl_and_the_literal_form(case):
    want, literal, contract, h = case
    well = h >= WELL
    assert rel_err(contract, want)[well].max() <= 1e-5
    assert rel_err(contract, literal)[well].max() <= 1e-5
    assert (rel_err(contract, want) <= 1e-5 + BOUND_EPS / h).all()
    # BASELINE's tolerance, per channel, on every sample (ill-conditioned ones included)
    assert (np.abs(contract - literal) <= 1e-4 * np.maximum(1.0, np.abs(literal))).all()


def test_the_alpha_channel_and_the_light_count_limits():
    frame, view, surf, args64 = random_case(7, 64, n_lights=1, types=(0,))
    out = bbo.light_surface(frame, view, surf, literal=False)
    assert (out[:, 3] == 1.0).all()  # outColor = vec4(color, 1), forward_brdf.frag:75
    frame["num_lights"] = 0  # only the ambient term: vec3(0.03) * albedo * ao (:73)
    amb = bbo.light_surface(frame, view, surf, literal=True)[:, :3]
    want = 0.03 * args64[4] * args64[7][:, None]
    assert np.abs(amb - want).max() <= 1e-7


@pytest.mark.parametrize("cfg,size", [(configs.C2, (192, 108)), (configs.C3, (240, 135))])
def test_contract_frames_against_literal_frames(cfg, size):
    """whole frames, texture sampling and interpolation included: BASELINE's tolerance between the two forms"""
    from bibim_renderer_amd import textures
    sc = scenes.shaderball_scene(cfg.scaled(size[0], size[1], 64), bbo.MaterialData(textures.make_material(64)))
    contract, prim, depth, _ = bbo.render(sc)
    literal, lprim, ldepth, _ = bbo.render(sc, flags=bbo.FLAG_LITERAL)
    assert np.array_equal(prim, lprim) and np.array_equal(depth.view(np.uint32), ldepth.view(np.uint32))
    assert (np.abs(contract - literal) <= 1e-4 * np.maximum(1.0, np.abs(literal))).all()
    assert np.abs(contract - literal).max() <= 2e-6 * max(1.0, float(np.abs(literal).max()))  # in fact much closer


@pytest.mark.parametrize("name", ["c2", "c3", "c5"])
def test_contract_frames_against_literal_frames_at_baseline_sizes(name):
    """The same comparison at the FULL size of BASELINE's GPU configurations with their 2048^2 maps (band-parallel, a few
    seconds): the GGX denominator only cancels to ~5e-4 on the smoothest texels of the full-size frames, so this is where
    a re-association would show.  Tolerance, stated: ABSOLUTE 1e-4 per channel (BASELINE.json: "within 1e-4 per
    channel"), which is stricter than BASELINE.md's 1e-4 * max(1, |ref|) wherever the HDR value exceeds 1 (up to 189 at
    C3); measured 2.3e-5."""
    from bibim_renderer_amd import textures
    cfg = configs.CONFIGS[name]
    sc = scenes.shaderball_scene(cfg, bbo.MaterialData(textures.make_material(cfg.texture_size)))
    contract, n_c = bbo.render_bands(sc)
    literal, n_l = bbo.render_bands(sc, flags=bbo.FLAG_LITERAL)
    assert n_c == n_l and np.isfinite(literal).all() and np.isfinite(contract).all()
    d = np.abs(contract.astype(np.float64) - literal.astype(np.float64))
    assert d.max() <= 1e-4, d.max()
    assert (d <= 1e-4 * np.maximum(1.0, np.abs(literal))).all()
    assert np.array_equal(contract[..., 3], literal[..., 3])   # alpha: 1 on geometry, 0 on the clear colour, both forms


# ---------------------------------------------------------------------------------------------------------------------
# texture sampling
# ---------------------------------------------------------------------------------------------------------------------
def np_bilinear(img, u, v):
    """VK_FILTER_LINEAR, VK_SAMPLER_ADDRESS_MODE_REPEAT, unnormalizedCoordinates = false on an R8G8B8A8_UNORM image
    (the reference's sampler: src/render.cpp createSampler, SMP_LINEAR).  Vulkan 1.3 spec 16.5-16.8: (s, t) * size - 0.5,
    i0 = floor, weights = frac, wrap = mod size; UNORM: texel / 255.  binary64."""
    h, w = img.shape[:2]
    x = np.asarray(u, np.float64) * w - 0.5
    y = np.asarray(v, np.float64) * h - 0.5
    i0, j0 = np.floor(x), np.floor(y)
    a, b = x - i0, y - j0
    i0, j0 = i0.astype(np.int64), j0.astype(np.int64)
    t = lambda i, j: img[np.mod(j, h), np.mod(i, w)].astype(np.float64) / 255.0
    return ((1 - a) * (1 - b))[:, None] * t(i0, j0) + (a * (1 - b))[:, None] * t(i0 + 1, j0) + \
        ((1 - a) * b)[:, None] * t(i0, j0 + 1) + (a * b)[:, None] * t(i0 + 1, j0 + 1)


@pytest.mark.parametrize("shape", [(64, 64), (17, 5), (1, 1), (48, 80), (2, 256)])
def test_sample_bilinear_against_numpy(shape):
    rng = np.random.Generator(np.random.PCG64(shape[0] * 1000 + shape[1]))
    img = rng.integers(0, 256, (shape[0], shape[1], 4), dtype=np.uint8)
    n = 30_000
    u = rng.uniform(-3, 4, n).astype(np.float32)
    v = rng.uniform(-3, 4, n).astype(np.float32)
    # texel centres and texel edges exactly (weights 0 and the floor boundary)
    u[:200] = ((rng.integers(-2 * shape[1], 3 * shape[1], 200) + 0.5) / shape[1]).astype(np.float32)
    v[:200] = (rng.integers(-2 * shape[0], 3 * shape[0], 200) / shape[0]).astype(np.float32)
    got = np.stack([bbo.sample(img, 0, float(a), float(b)) for a, b in zip(u, v)])
    want = np_bilinear(img, u, v)
    # binary32: u * size - 0.5 carries |u| size eps32 of coordinate error, i.e. up to that times 1 (full-range texel step)
    tol = 4 * np.float64(np.finfo(np.float32).eps) * (1.0 + np.maximum(np.abs(u) * shape[1], np.abs(v) * shape[0]))
    err = np.abs(got - want).max(1)
    on_edge = np.zeros(n, bool)
    # exactly on a floor boundary the binary32 coordinate may fall on the other side: the weight is then ~0 or ~1 and the
    # filtered value the same up to the same coordinate error -- no exclusion needed, the tolerance above covers it
    assert (err <= tol + 1e-7)[~on_edge].all(), (float(err.max()), int(np.argmax(err)))


def test_missing_maps_sample_the_default_material():
    """NULL image => resources/pbr/default/*.png (uniform images, read as data: one texel each)"""
    want = {0: (1, 1, 1), 1: (0, 0, 0), 2: (0, 0, 0), 3: (1, 1, 1), 4: (127 / 255, 127 / 255, 1), 5: (0, 0, 0)}
    for m, rgb in want.items():
        got = bbo.sample(None, m, 0.3, 0.7)
        assert np.allclose(got[:3], rgb, atol=1e-7), (m, got)


# ---------------------------------------------------------------------------------------------------------------------
# vertex stage and the head of the fragment stage
# ---------------------------------------------------------------------------------------------------------------------
def glsl_f64_vertex_stage(view, inst, vertex):
    """forward_brdf.vert:24-37 in binary64.  Matrices arrive as the reference stores them (Mat4::M[column][row], the
    bytes GLSL reads as column-major): numpy's [i][j] is column i, row j, so `M * v` is v @ M."""
    f = lambda a: np.asarray(a, np.float64)
    model, inv_model = f(inst["model"]), f(inst["inv_model"])
    pos = np.append(f(vertex["pos"]), 1.0)
    pos_world = pos @ model                                   # aModel * vec4(aPosition, 1.0)
    clip = (pos_world @ f(view["view"])) @ f(view["proj"])    # uProjMat * uViewMat * posWorld
    normal_mat = inv_model[:3, :3].T                          # transpose(mat3(aInvModel)), still [column][row]
    N = _normalize(f(vertex["normal"]) @ normal_mat)
    T = _normalize(f(vertex["tangent"]) @ normal_mat)
    B = np.cross(N, T)
    return clip, np.concatenate([f(vertex["uv"]), pos_world[:3], N, T, B])


def test_vertex_stage_against_the_glsl_in_binary64():
    rng = np.random.Generator(np.random.PCG64(77))
    view = scenes.view_uniforms((0.3, 2.0, -2.5), 20.0, -15.0, 1920, 1080, 1)
    worst = 0.0
    for _ in range(2000):
        # a rotation * non-uniform scale * translation, as the scene's instances are built (src/scene.cpp:180-187)
        m = bbo.mat_mul(bbo.mat_translate(*rng.uniform(-5, 5, 3)),
                        bbo.mat_mul(bbo.mat_rotate_y(rng.uniform(-180, 180)),
                                    bbo.mat_mul(bbo.mat_rotate_x(rng.uniform(-180, 180)), bbo.mat_scale(*rng.uniform(0.01, 2, 3)))))
        inst = np.zeros((), bbo.INSTANCE_DTYPE)
        inst["model"], inst["inv_model"] = m, bbo.mat_inverse(m)
        vtx = np.zeros((), bbo.VERTEX_DTYPE)
        vtx["pos"], vtx["uv"] = rng.uniform(-50, 50, 3), rng.uniform(0, 1, 2)
        n = rng.normal(size=3)
        t = np.cross(n, rng.normal(size=3))
        vtx["normal"], vtx["tangent"] = n / np.linalg.norm(n), t / np.linalg.norm(t)
        clip, vary = bbo.vertex_stage(view, inst, vtx)
        want_clip, want_vary = glsl_f64_vertex_stage(view, inst, vtx)
        worst = max(worst, np.abs(clip - want_clip).max() / np.abs(want_clip).max(),
                    np.abs(vary[:5] - want_vary[:5]).max() / max(1.0, np.abs(want_vary[:5]).max()),
                    np.abs(vary[5:] - want_vary[5:]).max())        # unit vectors: absolute
    assert worst <= 1e-5, worst


def test_fragment_head_normal_mapping_against_the_glsl_in_binary64():
    """forward_brdf.frag:16-25: the five texture reads and `normal = vTBN * (texture(normal).xyz * 2 - 1)`, then one light:
    the whole fragment stage through bbo.shade_fragment against sampling (np_bilinear) + the binary64 light loop"""
    from bibim_renderer_amd import textures
    rng = np.random.Generator(np.random.PCG64(91))
    maps = textures.make_material(32)
    mat = bbo.MaterialData(maps)
    lights = [scenes.light(0, pos=(1.0, 4.0, 2.0), color=(1.0, 0.8, 0.8), intensity=50.0)]
    frame = scenes.frame_uniforms(lights)
    view = scenes.view_uniforms((0.0, 3.0, 5.0), 0.0, 0.0, 64, 64, 1)
    l64 = [dict(type=0, pos=np.array([1.0, 4.0, 2.0]), dir=np.zeros(3), color=np.float64(np.float32([1.0, 0.8, 0.8])),
                intensity=50.0, inner=0.0, outer=0.0)]
    worst = 0.0
    n_checked = 0
    for _ in range(3000):
        uv = rng.uniform(-1, 2, 2).astype(np.float32)
        P = rng.uniform(-2, 2, 3).astype(np.float32)
        N = _normalize(rng.normal(size=3)); T = _normalize(np.cross(N, rng.normal(size=3))); B = np.cross(N, T)
        vary = np.concatenate([uv, P, N, T, B]).astype(np.float32)
        got = bbo.shade_fragment(frame, view, mat, vary, literal=True)[:3].astype(np.float64)
        s = lambda name: np_bilinear(maps[name], np.asarray(uv[:1], np.float64), np.asarray(uv[1:], np.float64))[0]
        albedo, metallic, roughness, ao = s("albedo")[:3], s("metallic")[0], s("roughness")[0], s("ao")[0]
        v64 = np.float64(vary)
        tbn = np.stack([v64[8:11], v64[11:14], v64[5:8]], axis=1)          # mat3(T, B, N): columns
        normal = tbn @ (s("normal")[:3] * 2 - 1)
        args = (l64, np.float64(np.float32([0.0, 3.0, 5.0])), v64[None, 2:5], normal[None], albedo[None],
                np.array([metallic]), np.array([roughness]), np.array([ao]))
        if conditioning(args[0], args[1], args[2], args[3], args[6])[0] < WELL:
            continue
        want = glsl_f64_light_loop(*args)[0]
        worst = max(worst, np.abs(got - want).max() / max(np.abs(want).max(), 1e-30))
        n_checked += 1
    assert n_checked >= 1500 and worst <= 2e-5, (n_checked, worst)   # (two binary32 stages in a row: sampling, then the loop)
