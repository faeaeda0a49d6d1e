"""Python handle on the C++ host shim (include/bibim_scene.h): ShaderBallScene / TriangleScene /
FreeLookCamera / draw_frame, i.e. the reference's Scene/Camera/"draw a frame" surface for the forward path.
All arithmetic happens in libbibim_hip.so (bb_scene.cpp); this file only marshals arguments."""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass

import numpy as np

from ._capi import BibimError, lib
from .renderer import Renderer

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

LIGHT_DTYPE = np.dtype(
    [("pos", "<f4", 3), ("type", "<i4"), ("dir", "<f4", 3), ("intensity", "<f4"), ("color", "<f4", 3),
     ("inner_cutoff", "<f4"), ("outer_cutoff", "<f4"), ("_pad", "<f4", 3)])
assert LIGHT_DTYPE.itemsize == 64


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def load_shaderball_vertices():
    """bb::Vertex[29328] float32 [n, 11]: the committed conversion of ShaderBall.fbx -- package data (its sha256 is pinned by
    tests/golden/shaderball_vertices.json; minted by tools/make_fixtures.py from the reference's asset)."""
    return np.ascontiguousarray(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "shaderball_vertices.npz"))["vertices"])


@dataclass
class FreeLookCamera:
    pos: tuple = (0.0, 0.0, 0.0)
    yaw: float = 0.0
    pitch: float = 0.0

    def view_matrix(self):
        out = np.zeros((4, 4), np.float32)
        lib().bbs_camera_view(_p(np.asarray(self.pos, np.float32)), self.yaw, self.pitch, _p(out))
        return out

    def look(self):
        out = np.zeros(3, np.float32)
        lib().bbs_camera_look(self.yaw, self.pitch, _p(out))
        return out


@dataclass
class FrameSettings:
    enable_normal_map: int = 0
    enable_tone_mapping: int = 0
    exposure: float = 1.0
    fov: float = 60.0
    near: float = 0.1
    far: float = 1000.0


class _Scene:
    def __init__(self, handle, renderer):
        if not handle:
            raise BibimError(-1, "scene creation failed")
        self._h = C.c_void_p(handle)
        self._r = renderer  # keeps the context alive
        if renderer is not None:
            renderer._scenes.add(self)  # the renderer closes its scenes before destroying the context

    def close(self):
        if self._h:
            lib().bbs_scene_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_lights(self, lights):
        a = np.ascontiguousarray(lights)
        assert a.dtype.itemsize == 64
        rc = lib().bbs_scene_set_lights(self._h, _p(a), a.size)
        if rc:
            raise BibimError(rc, "set_lights")

    def set_render_pass(self, deferred):
        """SceneBase::SceneRenderPassType: False = forward (this shim's default), True = deferred (the reference's)"""
        rc = lib().bbs_scene_set_render_pass(self._h, 1 if deferred else 0)
        if rc:
            raise BibimError(rc, "set_render_pass")

    def set_point_lights(self, point_lights):
        a = np.zeros(len(point_lights), LIGHT_DTYPE)
        for i, l in enumerate(point_lights):
            a[i]["pos"], a[i]["color"], a[i]["intensity"], a[i]["type"] = l.pos, l.color, l.intensity, 0
        self.set_lights(a)

    def lights(self):
        n = lib().bbs_scene_num_lights(self._h)
        a = np.zeros(n, LIGHT_DTYPE)
        if n:
            lib().bbs_scene_get_lights(self._h, _p(a))
        return a

    def instances(self, draw_index):
        n = C.c_uint32()
        rc = lib().bbs_scene_instances(self._h, draw_index, None, 0, C.byref(n))
        if rc:
            raise BibimError(rc, "instances")
        out = np.zeros((n.value, 2, 4, 4), np.float32)
        rc = lib().bbs_scene_instances(self._h, draw_index, _p(out), n.value, C.byref(n))
        if rc:
            raise BibimError(rc, "instances")
        return out

    def fill_uniforms(self, cam: FreeLookCamera, settings: FrameSettings, width, height):
        fb = np.zeros(6432, np.uint8)
        vb = np.zeros(144, np.uint8)
        rc = lib().bbs_fill_uniforms(self._h, _p(np.asarray(cam.pos, np.float32)), cam.yaw, cam.pitch,
                                     settings.enable_normal_map, settings.enable_tone_mapping, settings.exposure,
                                     settings.fov, settings.near, settings.far, width, height, _p(fb), _p(vb))
        if rc:
            raise BibimError(rc, "fill_uniforms")
        return fb, vb


class ShaderBallScene(_Scene):
    def __init__(self, renderer: Renderer | None, ball_vertices=None, grid=1, fbx_path=None):
        ctx = renderer._ctx if renderer is not None else None
        if fbx_path is not None:  # import the ball like the reference's constructor does (src/scene.cpp:57-86)
            super().__init__(lib().bbs_shaderball_scene_create_from_file(ctx, str(fbx_path).encode(), grid), renderer)
            return
        v = load_shaderball_vertices() if ball_vertices is None else np.ascontiguousarray(ball_vertices)
        super().__init__(lib().bbs_shaderball_scene_create(ctx, _p(v), v.shape[0], grid), renderer)


class TriangleScene(_Scene):
    def __init__(self, renderer: Renderer | None):
        ctx = renderer._ctx if renderer is not None else None
        super().__init__(lib().bbs_triangle_scene_create(ctx), renderer)


def draw_frame(renderer: Renderer, scene: _Scene, cam: FreeLookCamera, settings: FrameSettings, material: int):
    """One iteration of the reference's render loop for the forward path (asynchronous)."""
    rc = lib().bbs_draw_frame(renderer._ctx, scene._h, _p(np.asarray(cam.pos, np.float32)), cam.yaw, cam.pitch,
                              settings.enable_normal_map, settings.enable_tone_mapping, settings.exposure, settings.fov,
                              settings.near, settings.far, material, renderer.width, renderer.height)
    renderer._check(rc)


def frame_call(renderer: Renderer, scene: _Scene, cam: FreeLookCamera, settings: FrameSettings, material: int):
    """draw_frame with its arguments marshalled ONCE: returns a zero-argument callable that submits one frame.  A 1080p frame is as
    long as the host's own work per frame (bench.py's `host` block); building a numpy array and a dozen ctypes values per call is
    harness overhead, not the renderer's.  The camera, settings and material are those at the time of the call to frame_call."""
    fn = lib().bbs_draw_frame
    pos = np.ascontiguousarray(np.asarray(cam.pos, np.float32))
    args = (renderer._ctx, scene._h, _p(pos), C.c_float(cam.yaw), C.c_float(cam.pitch), C.c_int32(settings.enable_normal_map),
            C.c_int32(settings.enable_tone_mapping), C.c_float(settings.exposure), C.c_float(settings.fov), C.c_float(settings.near),
            C.c_float(settings.far), C.c_int32(material), C.c_int32(renderer.width), C.c_int32(renderer.height))
    check = renderer._check

    def submit(_keep=pos):
        rc = fn(*args)
        if rc:
            check(rc)
    return submit


def config_scene(renderer: Renderer, cfg, ball_vertices=None):
    """ShaderBallScene + camera + settings of a bibim_renderer_amd.configs.Config."""
    scene = ShaderBallScene(renderer, ball_vertices, cfg.grid)
    scene.set_point_lights(cfg.lights)
    cam = FreeLookCamera(cfg.cam_pos, cfg.cam_yaw, cfg.cam_pitch)
    settings = FrameSettings(cfg.enable_normal_map, 0, 1.0, cfg.fov, cfg.near, cfg.far)
    return scene, cam, settings
