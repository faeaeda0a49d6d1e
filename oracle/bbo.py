"""ctypes binding of the CPU oracle (oracle/libbb_oracle.so) + the scene inputs of BASELINE configs.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg.  The shipped package (bibim_renderer_amd/) must never import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def _cpu_has_fma() -> bool:
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("flags"):
                    return " fma " in line + " "
    except OSError:
        pass
    return False


def build(force: bool = False) -> None:
    """(Re)build the oracle library with gcc.  Building the checker is not using it."""
    so = os.path.join(HERE, "libbb_oracle.so")
    src = os.path.join(HERE, "bb_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", HERE, "all"])


def _load():
    name = "libbb_oracle.so" if _cpu_has_fma() else "libbb_oracle_nofma.so"
    path = os.path.join(HERE, name)
    if not os.path.exists(path):
        build()
    return C.CDLL(path)


# ---- layouts (reference: src/render.h:96-99,112-126,310-334) ----
VERTEX_DTYPE = np.dtype([("pos", "<f4", 3), ("uv", "<f4", 2), ("normal", "<f4", 3), ("tangent", "<f4", 3)])
GIZMO_VERTEX_DTYPE = np.dtype([("pos", "<f4", 3), ("color", "<f4", 3), ("normal", "<f4", 3)])
INSTANCE_DTYPE = np.dtype([("model", "<f4", (4, 4)), ("inv_model", "<f4", (4, 4))])
LIGHT_DTYPE = np.dtype(
    [("pos", "<f4", 3), ("type", "<i4"), ("dir", "<f4", 3), ("intensity", "<f4"), ("color", "<f4", 3),
     ("inner_cutoff", "<f4"), ("outer_cutoff", "<f4"), ("_pad", "<f4", 3)])
FRAME_DTYPE = np.dtype(
    [("num_lights", "<i4"), ("_pad0", "<i4", 3), ("lights", LIGHT_DTYPE, 100),
     ("visualized_gbuffer_attachment_index", "<i4"), ("enable_tone_mapping", "<i4"), ("exposure", "<f4"),
     ("_pad1", "<i4")])
VIEW_DTYPE = np.dtype([("view", "<f4", (4, 4)), ("proj", "<f4", (4, 4)), ("view_pos", "<f4", 3),
                       ("enable_normal_map", "<i4")])
assert VERTEX_DTYPE.itemsize == 44 and INSTANCE_DTYPE.itemsize == 128 and LIGHT_DTYPE.itemsize == 64
assert FRAME_DTYPE.itemsize == 6432 and VIEW_DTYPE.itemsize == 144 and GIZMO_VERTEX_DTYPE.itemsize == 36

MAP_NAMES = ("albedo", "metallic", "roughness", "ao", "normal", "height")  # PBRMapType, src/render.h:235-243


class Image(C.Structure):
    _fields_ = [("rgba", C.c_void_p), ("w", C.c_int32), ("h", C.c_int32)]


class Material(C.Structure):
    _fields_ = [("maps", Image * 6)]


class Draw(C.Structure):
    _fields_ = [("vertices", C.c_void_p), ("n_vertices", C.c_uint32), ("indices", C.c_void_p),
                ("n_indices", C.c_uint32), ("instances", C.c_void_p), ("n_instances", C.c_uint32),
                ("material", C.POINTER(Material))]


class Stats(C.Structure):
    _fields_ = [("n_prims", C.c_uint64), ("n_raster_tris", C.c_uint64), ("n_clipped_prims", C.c_uint64),
                ("n_fragments", C.c_uint64), ("n_shaded", C.c_uint64)]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


NO_PRIM = 0xFFFFFFFF
FLAG_FORWARD_SHADE = 1
FLAG_LITERAL = 4  # light loop statement by statement as the GLSL is written, instead of the shipped evaluation order
FLAG_OUTPUT_UV = 8  # measurement aid: the winning fragment's interpolated vUV instead of its colour

_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = _load()
        L = _lib
        L.bbo_render_parallel.restype = C.c_int
        L.bbo_render_parallel.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(Draw), C.c_uint32, C.c_int32, C.c_int32, C.c_uint32,
                                          C.c_int32, C.c_int32, C.c_void_p, C.POINTER(Stats)]
        L.bbo_render.restype = C.c_int
        L.bbo_render.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(Draw), C.c_uint32, C.c_int32, C.c_int32, C.c_int32,
                                 C.c_int32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(Stats)]
        L.bbo_render_gizmo.restype = C.c_int
        L.bbo_render_gizmo.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_int32,
                                       C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(Stats)]
        L.bbo_vertex_stage.argtypes = [C.c_void_p] * 5
        L.bbo_proj_view.argtypes = [C.c_void_p] * 2
        L.bbo_sample.argtypes = [C.POINTER(Image), C.c_int, C.c_float, C.c_float, C.c_void_p]
        L.bbo_shade_fragment.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(Material), C.c_void_p, C.c_void_p]
        L.bbo_shade_fragment_contract.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(Material), C.c_void_p, C.c_void_p]
        L.bbo_light_surface.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.bbo_light_surface.restype = None
        L.bbo_distribution_ggx.restype = C.c_float
        L.bbo_distribution_ggx.argtypes = [C.c_void_p, C.c_void_p, C.c_float]
        L.bbo_geometry_smith.restype = C.c_float
        L.bbo_geometry_smith.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float]
        L.bbo_fresnel_schlick.argtypes = [C.c_void_p] * 4
        L.bbo_tone_map.argtypes = [C.c_void_p, C.c_uint64, C.c_int32, C.c_float]
        L.bbo_present.argtypes = [C.c_void_p, C.c_uint64, C.c_int32, C.c_float, C.c_int32, C.c_void_p]
        L.bbo_present.restype = None
        L.bbo_half_round.argtypes = [C.c_float]
        L.bbo_half_round.restype = C.c_float
        L.bbo_half_round_n.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.bbo_half_round_n.restype = None
        L.bbo_exp.argtypes = [C.c_float]
        L.bbo_exp.restype = C.c_float
        L.bbo_srgb_thresholds.argtypes = [C.c_void_p]
        L.bbo_srgb_thresholds.restype = None
        for n in ("identity",):
            getattr(L, f"bbo_mat4_{n}").argtypes = [C.c_void_p]
        L.bbo_mat4_mul.argtypes = [C.c_void_p] * 3
        L.bbo_mat4_inverse.argtypes = [C.c_void_p] * 2
        L.bbo_mat4_transpose.argtypes = [C.c_void_p] * 2
        L.bbo_mat4_translate.argtypes = [C.c_float] * 3 + [C.c_void_p]
        L.bbo_mat4_scale.argtypes = [C.c_float] * 3 + [C.c_void_p]
        for n in ("rotate_x", "rotate_y", "rotate_z"):
            getattr(L, f"bbo_mat4_{n}").argtypes = [C.c_float, C.c_void_p]
        L.bbo_mat4_look_at.argtypes = [C.c_void_p] * 4
        L.bbo_mat4_perspective.argtypes = [C.c_float] * 4 + [C.c_void_p]
        L.bbo_camera_look.argtypes = [C.c_float, C.c_float, C.c_void_p]
        L.bbo_camera_view.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_void_p]
        L.bbo_sizeof.restype = C.c_uint32
        L.bbo_sizeof.argtypes = [C.c_int]
        L.bbo_contract_revision.restype = C.c_uint32
        L.bbo_contract_revision.argtypes = []
    return _lib


def contract_revision() -> int:
    """BBO_CONTRACT_REVISION of the built oracle: which evaluation order the default (shipped) form is (bb_oracle.h)."""
    return int(lib().bbo_contract_revision())


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _f3(v):
    return np.ascontiguousarray(v, dtype=np.float32)


# ---- matrix helpers (row A0) ----
def _m():
    return np.zeros((4, 4), dtype=np.float32)


def mat_mul(a, b):
    o = _m(); lib().bbo_mat4_mul(_p(np.ascontiguousarray(a, np.float32)), _p(np.ascontiguousarray(b, np.float32)), _p(o)); return o


def mat_inverse(a):
    o = _m(); lib().bbo_mat4_inverse(_p(np.ascontiguousarray(a, np.float32)), _p(o)); return o


def mat_transpose(a):
    o = _m(); lib().bbo_mat4_transpose(_p(np.ascontiguousarray(a, np.float32)), _p(o)); return o


def mat_translate(x, y, z):
    o = _m(); lib().bbo_mat4_translate(x, y, z, _p(o)); return o


def mat_scale(x, y=None, z=None):
    if y is None:
        y = z = x
    o = _m(); lib().bbo_mat4_scale(x, y, z, _p(o)); return o


def mat_rotate_x(d):
    o = _m(); lib().bbo_mat4_rotate_x(d, _p(o)); return o


def mat_rotate_y(d):
    o = _m(); lib().bbo_mat4_rotate_y(d, _p(o)); return o


def mat_rotate_z(d):
    o = _m(); lib().bbo_mat4_rotate_z(d, _p(o)); return o


def mat_look_at(eye, target, up=(0, 1, 0)):
    o = _m(); lib().bbo_mat4_look_at(_p(_f3(eye)), _p(_f3(target)), _p(_f3(up)), _p(o)); return o


def mat_perspective(fov, aspect, n, f):
    o = _m(); lib().bbo_mat4_perspective(fov, aspect, n, f, _p(o)); return o


def camera_look(yaw, pitch):
    o = np.zeros(3, np.float32); lib().bbo_camera_look(yaw, pitch, _p(o)); return o


def camera_view(pos, yaw, pitch):
    o = _m(); lib().bbo_camera_view(_p(_f3(pos)), yaw, pitch, _p(o)); return o


# ---- scene containers ----
class MaterialData:
    """Six optional RGBA8 maps (numpy uint8 [h, w, 4]); None => `default` material map."""

    def __init__(self, maps: dict | None = None):
        self.maps = {}
        for k, v in (maps or {}).items():
            if k not in MAP_NAMES:
                raise KeyError(k)
            if v is not None:
                v = np.ascontiguousarray(v, dtype=np.uint8)
                assert v.ndim == 3 and v.shape[2] == 4
                self.maps[k] = v
        self._c = Material()
        for i, name in enumerate(MAP_NAMES):
            a = self.maps.get(name)
            if a is None:
                self._c.maps[i] = Image(None, 0, 0)
            else:
                self._c.maps[i] = Image(a.ctypes.data, a.shape[1], a.shape[0])


class DrawData:
    def __init__(self, vertices, indices, instances, material: MaterialData):
        self.vertices = np.ascontiguousarray(vertices)
        assert self.vertices.dtype == VERTEX_DTYPE
        self.indices = None if indices is None else np.ascontiguousarray(indices, dtype=np.uint32)
        self.instances = np.ascontiguousarray(instances)
        assert self.instances.dtype == INSTANCE_DTYPE
        self.material = material

    @property
    def n_prims(self):
        n = len(self.indices) if self.indices is not None else len(self.vertices)
        return (n // 3) * len(self.instances)

    def c_struct(self):
        return Draw(self.vertices.ctypes.data, len(self.vertices),
                    self.indices.ctypes.data if self.indices is not None else None,
                    len(self.indices) if self.indices is not None else 0,
                    self.instances.ctypes.data, len(self.instances), C.pointer(self.material._c))


class Scene:
    """Everything one frame consumes: uniforms + draws in API order + target size."""

    def __init__(self, frame, view, draws, width, height, name=""):
        self.frame = frame  # np.ndarray shape () of FRAME_DTYPE
        self.view = view
        self.draws = list(draws)
        self.width, self.height = int(width), int(height)
        self.name = name

    @property
    def n_prims(self):
        return sum(d.n_prims for d in self.draws)


def render(scene: Scene, y0=0, y1=None, flags=0, want_prim=True, want_depth=True):
    """Oracle render of rows [y0, y1).  Returns (rgba[h,w,4] f32, prim[h,w] u32|None, depth|None, stats dict)."""
    W, H = scene.width, scene.height
    if y1 is None:
        y1 = H
    rgba = np.zeros((H, W, 4), np.float32)
    prim = np.full((H, W), NO_PRIM, np.uint32) if want_prim else None
    depth = np.zeros((H, W), np.float32) if want_depth else None
    arr = (Draw * max(1, len(scene.draws)))(*[d.c_struct() for d in scene.draws])
    st = Stats()
    rc = lib().bbo_render(_p(scene.frame), _p(scene.view), arr, len(scene.draws), W, H, y0, y1, flags, _p(rgba),
                          _p(prim), _p(depth), C.byref(st))
    if rc != 0:
        raise RuntimeError(f"bbo_render failed: {rc}")
    return rgba, prim, depth, st.as_dict()


def render_bands(scene: Scene, flags=0, rows=32, threads=None):
    """The whole frame, bands of `rows` rows over the host's cores (the C call drops the GIL; bands share nothing, so the
    bits are those of one whole-frame call).  For BASELINE's full sizes: 4K in ~1 s, 8K in a few.  Returns (rgba, n_shaded)."""
    from concurrent.futures import ThreadPoolExecutor
    W, H = scene.width, scene.height
    rgba = np.zeros((H, W, 4), np.float32)
    arr = (Draw * max(1, len(scene.draws)))(*[d.c_struct() for d in scene.draws])
    L = lib()

    def band(y0):
        st = Stats()
        rc = L.bbo_render(_p(scene.frame), _p(scene.view), arr, len(scene.draws), W, H, y0, min(y0 + rows, H), flags, _p(rgba),
                          None, None, C.byref(st))
        if rc != 0:
            raise RuntimeError(f"bbo_render failed: {rc}")
        return st.n_shaded

    if threads is None:
        threads = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    with ThreadPoolExecutor(max(1, min(threads, 16))) as ex:
        n = sum(ex.map(band, range(0, H, rows)))
    return rgba, int(n)


def render_parallel(scene: Scene, flags=0, threads=None, rows=8, out=None):
    """The whole forward frame on `threads` threads inside the C library (primitives set up once, bands from a queue): bit for bit
    render()'s frame.  Returns (rgba, stats dict).  What bench.py's all-cores CPU baseline times."""
    W, H = scene.width, scene.height
    rgba = np.zeros((H, W, 4), np.float32) if out is None else out
    arr = (Draw * max(1, len(scene.draws)))(*[d.c_struct() for d in scene.draws])
    if threads is None:
        threads = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    st = Stats()
    rc = lib().bbo_render_parallel(_p(scene.frame), _p(scene.view), arr, len(scene.draws), W, H, flags, int(threads), int(rows),
                                   _p(rgba), C.byref(st))
    if rc != 0:
        raise RuntimeError(f"bbo_render_parallel failed: {rc}")
    return rgba, st.as_dict()


def render_deferred(scene: Scene, y0=0, y1=None, want_gbuffer=True, flags=0):
    """The reference's deferred path (gbuffer.vert/.frag + brdf.frag).  Returns (rgba[h,w,4], gbuffer[h,w,4,4]|None,
    prim, depth, stats); gbuffer[..., a, :] = attachment a (position, normal, albedo, MRAH), binary16-representable."""
    W, H = scene.width, scene.height
    if y1 is None:
        y1 = H
    rgba = np.zeros((H, W, 4), np.float32)
    gbuf = np.zeros((H, W, 4, 4), np.float32) if want_gbuffer else None
    prim = np.full((H, W), NO_PRIM, np.uint32)
    depth = np.zeros((H, W), np.float32)
    arr = (Draw * max(1, len(scene.draws)))(*[d.c_struct() for d in scene.draws])
    st = Stats()
    L = lib()
    L.bbo_render_deferred_flags.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int32, C.c_int32, C.c_int32,
                                            C.c_int32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    rc = L.bbo_render_deferred_flags(_p(scene.frame), _p(scene.view), arr, len(scene.draws), W, H, y0, y1, flags, _p(rgba),
                                     _p(gbuf), _p(prim), _p(depth), C.byref(st))
    if rc != 0:
        raise RuntimeError(f"bbo_render_deferred failed: {rc}")
    return rgba, gbuf, prim, depth, st.as_dict()


def uv_sphere(radius=0.1, hdiv=16, vdiv=16):
    """generateUVSphereMesh positions [n, 3] f32 and indices [m] u32 (the light-marker mesh)"""
    L = lib()
    L.bbo_uv_sphere.argtypes = [C.c_float, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.bbo_uv_sphere.restype = None
    nv, ni = C.c_uint32(), C.c_uint32()
    L.bbo_uv_sphere(radius, hdiv, vdiv, None, None, C.byref(nv), C.byref(ni))
    pos = np.zeros((nv.value, 3), np.float32)
    idx = np.zeros(ni.value, np.uint32)
    L.bbo_uv_sphere(radius, hdiv, vdiv, _p(pos), _p(idx), None, None)
    return pos, idx


def overlay(frame, view, scene_depth, rgba8, gizmo_vertices=None, gizmo_indices=None, gizmo_extent=100):
    """light markers + corner gizmo over a presented RGBA8 image (returns a new array), depth-tested against scene_depth"""
    L = lib()
    L.bbo_overlay.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32,
                              C.c_void_p, C.c_uint32, C.c_int32, C.c_void_p]
    out = np.ascontiguousarray(rgba8, np.uint8).copy()
    H, W = out.shape[:2]
    depth = np.ascontiguousarray(scene_depth, np.float32)
    assert depth.shape == (H, W)
    gv = None if gizmo_vertices is None else np.ascontiguousarray(gizmo_vertices)
    gi = None if gizmo_indices is None else np.ascontiguousarray(gizmo_indices, np.uint32)
    if gv is not None:
        assert gv.dtype == GIZMO_VERTEX_DTYPE
    st = Stats()
    rc = L.bbo_overlay(_p(frame), _p(view), W, H, _p(depth), _p(out), _p(gv), 0 if gv is None else len(gv), _p(gi),
                       0 if gi is None else len(gi), gizmo_extent, C.byref(st))
    if rc != 0:
        raise RuntimeError(f"bbo_overlay failed: {rc}")
    return out, st.as_dict()


def render_gizmo(view, vertices, indices, width, height):
    rgba = np.zeros((height, width, 4), np.float32)
    prim = np.full((height, width), NO_PRIM, np.uint32)
    depth = np.zeros((height, width), np.float32)
    st = Stats()
    vertices = np.ascontiguousarray(vertices)
    assert vertices.dtype == GIZMO_VERTEX_DTYPE
    idx = None if indices is None else np.ascontiguousarray(indices, np.uint32)
    rc = lib().bbo_render_gizmo(_p(view), _p(vertices), len(vertices), _p(idx), 0 if idx is None else len(idx), width,
                                height, _p(rgba), _p(prim), _p(depth), C.byref(st))
    if rc != 0:
        raise RuntimeError(f"bbo_render_gizmo failed: {rc}")
    return rgba, prim, depth, st.as_dict()


def vertex_stage(view, inst, vertex):
    clip = np.zeros(4, np.float32); vary = np.zeros(14, np.float32)
    lib().bbo_vertex_stage(_p(view), _p(inst), _p(vertex), _p(clip), _p(vary))
    return clip, vary


def sample(img, map_type, u, v):
    out = np.zeros(4, np.float32)
    if img is None:
        im = Image(None, 0, 0)
    else:
        img = np.ascontiguousarray(img, np.uint8)
        im = Image(img.ctypes.data, img.shape[1], img.shape[0])
    lib().bbo_sample(C.byref(im), map_type, u, v, _p(out))
    return out


def shade_fragment(frame, view, material: MaterialData, vary, literal=True):
    """forward_brdf.frag on explicit varyings: statement by statement (literal) or in the shipped evaluation order"""
    out = np.zeros(4, np.float32)
    vary = np.ascontiguousarray(vary, np.float32)
    fn = lib().bbo_shade_fragment if literal else lib().bbo_shade_fragment_contract
    fn(_p(frame), _p(view), C.byref(material._c), _p(vary), _p(out))
    return out


def light_surface(frame, view, surf, literal=True):
    """The light loop + ambient term on surface points surf[n, 12] = P(3) normal(3) albedo(3) metallic roughness ao."""
    surf = np.ascontiguousarray(surf, np.float32).reshape(-1, 12)
    out = np.zeros((len(surf), 4), np.float32)
    fn = lib().bbo_light_surface
    fp, vp = _p(frame), _p(view)
    for i in range(len(surf)):
        fn(int(bool(literal)), fp, vp, surf[i].ctypes.data, out[i].ctypes.data)
    return out


def tone_map(rgba, enable, exposure):
    out = np.ascontiguousarray(rgba, np.float32).copy()
    lib().bbo_tone_map(_p(out), out.size // 4, int(enable), float(exposure))
    return out


def present(rgba, enable, exposure, hdr16=True):
    """fp32 RGBA frame -> presented RGBA8 (binary16 HDR attachment, tone map, sRGB, UNORM8)"""
    src = np.ascontiguousarray(rgba, np.float32)
    out = np.empty(src.shape, np.uint8)
    lib().bbo_present(_p(src), src.size // 4, int(enable), float(exposure), int(bool(hdr16)), _p(out))
    return out


def half_round(x):
    """every value to the nearest binary16 value, ties to even (bbo_half_round, the rounding of an RGBA16F attachment write)"""
    a = np.ascontiguousarray(x, np.float32)
    if a.size > 64:
        out = np.empty_like(a)
        lib().bbo_half_round_n(_p(a), _p(out), a.size)
        return out.reshape(np.shape(x))
    return np.array([lib().bbo_half_round(float(v)) for v in np.ravel(a)], np.float32).reshape(np.shape(x))


def exp(x):
    return np.array([lib().bbo_exp(float(v)) for v in np.ravel(x)], np.float32).reshape(np.shape(x))


def srgb_thresholds():
    out = np.empty(255, np.float32)
    lib().bbo_srgb_thresholds(_p(out))
    return out
