"""Row A0: the oracle's restatement of vector_math.cpp / camera.cpp is bit-identical to the reference's own
sources -- via the committed golden vectors (minted from oracle/_ref) and, when oracle/_ref is present
(authoring container), live on random inputs."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, ROOT
from oracle import bbo


def _bits(a):
    return [int(x) for x in np.ascontiguousarray(a, np.float32).view(np.uint32).ravel()]


def _from_bits(b):
    return np.array(b, np.uint32).view(np.float32).reshape(4, 4)


def _eval_case(m, c):
    op, a = c["op"], c["args"]
    if op == "perspective":
        return m.mat_perspective(*a)
    if op.startswith("rotate_"):
        return getattr(m, "mat_" + op)(a[0])
    if op == "camera_look":
        return m.camera_look(*a)
    if op == "camera_view":
        return m.camera_view(a[0], a[1], a[2])
    if op == "look_at":
        return m.mat_look_at(*a)
    if op == "mul":
        return m.mat_mul(_from_bits(a[0]), _from_bits(a[1]))
    if op == "inverse":
        return m.mat_inverse(_from_bits(a[0]))
    if op in ("instance_chain", "instance_chain_inverse"):
        t = m.mat_translate(*a)
        chain = m.mat_mul(m.mat_mul(m.mat_mul(t, m.mat_rotate_y(-90.0)), m.mat_rotate_x(-90.0)), m.mat_scale(0.01, 0.01, 0.01))
        return chain if op == "instance_chain" else m.mat_inverse(chain)
    if op in ("plane_model", "plane_model_inverse"):
        pm = m.mat_mul(m.mat_translate(0.0, -10.0, 0.0), m.mat_scale(100.0, 100.0, 100.0))
        return pm if op == "plane_model" else m.mat_inverse(pm)
    raise KeyError(op)


def golden_cases():
    return json.load(open(os.path.join(GOLDEN, "math_golden.json")))["cases"]


def test_oracle_matches_reference_golden_vectors():
    cases = golden_cases()
    assert len(cases) >= 80
    for c in cases:
        assert _bits(_eval_case(bbo, c)) == c["bits"], (c["op"], c["args"])


def test_survey_sample_values():
    # perspective(60, 16/9, 0.1, 1000)
    p = bbo.mat_perspective(60.0, 16.0 / 9.0, 0.1, 1000.0)
    assert p[0, 0] == np.float32(0.974278808) and p[1, 1] == np.float32(-1.73205125)
    assert p[2, 2] == np.float32(-0.000100010002) and p[3, 2] == np.float32(0.10001) and p[2, 3] == 1.0
    # pi32 = 3.141592f makes cos(-90 deg) tiny but non-zero: Model.M[0][0] = 3.139e-09 after scale(0.01)
    m = _eval_case(bbo, {"op": "instance_chain", "args": [0.0, -1.0, 2.0]})
    assert m[0, 0] == np.float32(3.13916471e-09)


@pytest.mark.skipif(not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libbb_ref.so")),
                    reason="oracle/_ref is only built where /root/reference exists")
def test_oracle_matches_compiled_reference_live():
    ref = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libbb_ref.so"))
    P, f = C.c_void_p, C.c_float
    ref.ref_mat4_mul.argtypes = [P, P, P]; ref.ref_mat4_inverse.argtypes = [P, P]
    ref.ref_mat4_perspective.argtypes = [f, f, f, f, P]; ref.ref_camera_view.argtypes = [P, f, f, P]
    ref.ref_mat4_look_at.argtypes = [P, P, P, P]
    for n in "xyz":
        getattr(ref, f"ref_mat4_rotate_{n}").argtypes = [f, P]
    rng = np.random.Generator(np.random.PCG64(99))
    p = lambda a: a.ctypes.data_as(P)
    for _ in range(200):
        a = rng.standard_normal((4, 4)).astype(np.float32) * np.float32(10 ** rng.uniform(-2, 2))
        b = rng.standard_normal((4, 4)).astype(np.float32)
        o = np.zeros((4, 4), np.float32)
        ref.ref_mat4_mul(p(a), p(b), p(o)); assert _bits(o) == _bits(bbo.mat_mul(a, b))
        ref.ref_mat4_inverse(p(a), p(o)); assert _bits(o) == _bits(bbo.mat_inverse(a))
        d = float(rng.uniform(-720, 720))
        for n in "xyz":
            getattr(ref, f"ref_mat4_rotate_{n}")(d, p(o)); assert _bits(o) == _bits(getattr(bbo, f"mat_rotate_{n}")(d))
        fov, asp = float(rng.uniform(5, 170)), float(rng.uniform(0.2, 4))
        ref.ref_mat4_perspective(fov, asp, 0.1, 1000.0, p(o)); assert _bits(o) == _bits(bbo.mat_perspective(fov, asp, 0.1, 1000.0))
        pos = rng.standard_normal(3).astype(np.float32) * 5
        yaw, pitch = float(rng.uniform(-180, 180)), float(rng.uniform(-89, 89))
        ref.ref_camera_view(p(pos), yaw, pitch, p(o)); assert _bits(o) == _bits(bbo.camera_view(pos, yaw, pitch))
        eye, tgt = rng.standard_normal(3).astype(np.float32), rng.standard_normal(3).astype(np.float32) * 4
        up = np.array([0, 1, 0], np.float32)
        ref.ref_mat4_look_at(p(eye), p(tgt), p(up), p(o)); assert _bits(o) == _bits(bbo.mat_look_at(eye, tgt, up))


def test_layout_sizes():
    L = bbo.lib()
    assert [L.bbo_sizeof(i) for i in range(6)] == [44, 128, 64, 6432, 144, 36]


def test_default_material_texels_match_reference_files():
    """resources/pbr/default/*.png decoded by the reference's stb_image (fixture) == the oracle's built-in defaults."""
    d = json.load(open(os.path.join(GOLDEN, "default_texels.json")))["maps"]
    for i, name in enumerate(("albedo", "metallic", "roughness", "ao", "normal", "height")):
        assert d[name]["uniform"]
        got = bbo.sample(None, i, 0.3, 0.7)
        want = np.array(d[name]["texel"], np.float32) * np.float32(1.0 / 255.0)
        assert np.array_equal(got, want), name
