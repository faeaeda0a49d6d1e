"""The C++ Scene/Camera/drawFrame shim (bb_scene.cpp, the product's host side) reproduces the reference's math
bit-for-bit (golden vectors from oracle/_ref) and builds byte-identical frame inputs to the oracle-side scenes."""
import ctypes as C

import numpy as np

from bibim_renderer_amd import configs
from bibim_renderer_amd import scene as S
from bibim_renderer_amd._capi import lib
from oracle import bbo, scenes
from test_oracle_math import _bits, _eval_case, golden_cases


class ShimMath:
    """bbs_* wrapped with the oracle module's function names so _eval_case can drive either."""

    @staticmethod
    def _m():
        return np.zeros((4, 4), np.float32)

    @staticmethod
    def _p(a):
        return np.ascontiguousarray(a, np.float32).ctypes.data_as(C.c_void_p)

    def mat_mul(self, a, b):
        o = self._m(); a = np.ascontiguousarray(a, np.float32); b = np.ascontiguousarray(b, np.float32)
        lib().bbs_mat4_mul(a.ctypes.data, b.ctypes.data, o.ctypes.data); return o

    def mat_inverse(self, a):
        o = self._m(); a = np.ascontiguousarray(a, np.float32); lib().bbs_mat4_inverse(a.ctypes.data, o.ctypes.data); return o

    def mat_translate(self, x, y, z):
        o = self._m(); lib().bbs_mat4_translate(x, y, z, o.ctypes.data); return o

    def mat_scale(self, x, y, z):
        o = self._m(); lib().bbs_mat4_scale(x, y, z, o.ctypes.data); return o

    def mat_rotate_x(self, d):
        o = self._m(); lib().bbs_mat4_rotate(0, d, o.ctypes.data); return o

    def mat_rotate_y(self, d):
        o = self._m(); lib().bbs_mat4_rotate(1, d, o.ctypes.data); return o

    def mat_rotate_z(self, d):
        o = self._m(); lib().bbs_mat4_rotate(2, d, o.ctypes.data); return o

    def mat_perspective(self, fov, asp, n, f):
        o = self._m(); lib().bbs_mat4_perspective(fov, asp, n, f, o.ctypes.data); return o

    def mat_look_at(self, eye, tgt, up):
        o = self._m()
        e, t, u = (np.asarray(v, np.float32) for v in (eye, tgt, up))
        lib().bbs_mat4_look_at(e.ctypes.data, t.ctypes.data, u.ctypes.data, o.ctypes.data); return o

    def camera_look(self, yaw, pitch):
        o = np.zeros(3, np.float32); lib().bbs_camera_look(yaw, pitch, o.ctypes.data); return o

    def camera_view(self, pos, yaw, pitch):
        o = self._m(); p = np.asarray(pos, np.float32); lib().bbs_camera_view(p.ctypes.data, yaw, pitch, o.ctypes.data); return o


def test_shim_math_matches_reference_golden_vectors():
    m = ShimMath()
    for c in golden_cases():
        assert _bits(_eval_case(m, c)) == c["bits"], (c["op"], c["args"])


def test_plane_mesh_matches_generatePlaneMesh():
    v = np.zeros(4, bbo.VERTEX_DTYPE); i = np.zeros(6, np.uint32)
    lib().bbs_plane_mesh(v.ctypes.data, i.ctypes.data)
    ov, oi = scenes.plane_mesh()
    assert v.tobytes() == ov.tobytes() and i.tolist() == oi.tolist() == [0, 1, 2, 2, 3, 0]


def test_shaderball_scene_defaults_match_reference_constructor():
    sc = S.ShaderBallScene(None, grid=1)  # no context: host logic only
    lights = sc.lights()
    want = scenes.reference_default_lights()
    assert len(lights) == 3
    for got, w in zip(lights, want):
        assert got.tobytes() == w.tobytes()
    inst = sc.instances(0)
    assert inst.shape[0] == 1
    assert inst.tobytes() == scenes.ball_instances(1).tobytes()
    sc.close()


def test_config_scenes_build_identical_inputs():
    for cfg in (configs.C2, configs.C3, configs.C5):
        sc, cam, settings = S.config_scene(None, cfg)
        osc = scenes.shaderball_scene(cfg, bbo.MaterialData())
        fb, vb = sc.fill_uniforms(cam, settings, cfg.width, cfg.height)
        assert fb.tobytes() == osc.frame.tobytes(), cfg.name
        assert vb.tobytes() == osc.view.tobytes(), cfg.name
        assert sc.instances(0).tobytes() == osc.draws[0].instances.tobytes(), cfg.name
        assert sc.instances(1).tobytes() == osc.draws[1].instances.tobytes(), cfg.name
        sc.close()


def test_triangle_scene_inputs():
    sc = S.TriangleScene(None)
    osc = scenes.triangle_scene(64, 64)
    fb, vb = sc.fill_uniforms(S.FreeLookCamera(), S.FrameSettings(), 64, 64)
    assert fb.tobytes() == osc.frame.tobytes() and vb.tobytes() == osc.view.tobytes()
    assert sc.instances(0).tobytes() == osc.draws[0].instances.tobytes()
    sc.close()


def test_set_lights_rejects_100():
    sc = S.ShaderBallScene(None, grid=1)
    a = np.zeros(100, S.LIGHT_DTYPE)
    try:
        sc.set_lights(a)  # reference asserts NumLights < MAX_NUM_LIGHTS (src/main.cpp:1289-1290)
        assert False, "accepted 100 lights"
    except Exception as e:
        assert "INVALID" in str(e)
    sc.close()
