"""ctypes wrappers over include/bibim_assets.h (FBX / OBJ / PNG readers and the pbr/<name>/ directory convention)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _capi

_P = C.c_void_p
_SIGS = {
    "bba_last_error": (C.c_char_p, []),
    "bba_free": (None, [_P]),
    "bba_load_fbx_vertices": (C.c_int, [C.c_char_p, C.POINTER(_P), C.POINTER(C.c_uint32)]),
    "bba_load_obj_gizmo": (C.c_int, [C.c_char_p, C.POINTER(_P), C.POINTER(C.c_uint32), C.POINTER(_P), C.POINTER(C.c_uint32)]),
    "bba_load_png": (C.c_int, [C.c_char_p, C.POINTER(_P), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "bba_decode_png": (C.c_int, [_P, C.c_uint64, C.POINTER(_P), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "bba_load_material_dir": (C.c_int, [_P, C.c_char_p, C.POINTER(C.c_int32)]),
    "bba_load_material_set": (C.c_int, [_P, C.c_char_p, _P, _P, C.c_uint32, C.POINTER(C.c_uint32)]),
}
_bound = False


class AssetError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"bba error {code}: {msg}")
        self.code = code


def _lib():
    global _bound
    L = _capi.lib()
    if not _bound:
        for name, (res, args) in _SIGS.items():
            f = getattr(L, name)
            f.restype, f.argtypes = res, args
        _bound = True
    return L


def _check(rc):
    if rc != 0:
        raise AssetError(rc, _lib().bba_last_error().decode("utf-8", "replace"))


def _take(ptr, nbytes, dtype, shape):
    """copy a malloc'ed buffer into numpy and free it"""
    L = _lib()
    try:
        buf = (C.c_uint8 * nbytes).from_address(ptr.value) if nbytes else b""
        return np.frombuffer(bytes(buf), dtype=dtype).reshape(shape).copy()
    finally:
        L.bba_free(ptr)


def load_fbx_vertices(path):
    """binary FBX -> float32 [n, 11] in bb::Vertex order (pos 3, uv 2, normal 3, tangent 3)"""
    p, n = _P(), C.c_uint32()
    _check(_lib().bba_load_fbx_vertices(str(path).encode(), C.byref(p), C.byref(n)))
    return _take(p, n.value * 44, np.float32, (n.value, 11))


def load_obj_gizmo(path):
    """OBJ/MTL -> (float32 [n, 9] = pos, colour, normal; uint32 [m] indices)"""
    pv, nv, pi, ni = _P(), C.c_uint32(), _P(), C.c_uint32()
    _check(_lib().bba_load_obj_gizmo(str(path).encode(), C.byref(pv), C.byref(nv), C.byref(pi), C.byref(ni)))
    v = _take(pv, nv.value * 36, np.float32, (nv.value, 9))
    i = _take(pi, ni.value * 4, np.uint32, (ni.value,))
    return v, i


def load_png(path):
    p, w, h = _P(), C.c_int32(), C.c_int32()
    _check(_lib().bba_load_png(str(path).encode(), C.byref(p), C.byref(w), C.byref(h)))
    return _take(p, w.value * h.value * 4, np.uint8, (h.value, w.value, 4))


def decode_png(data: bytes):
    p, w, h = _P(), C.c_int32(), C.c_int32()
    buf = (C.c_uint8 * len(data)).from_buffer_copy(data)
    _check(_lib().bba_decode_png(C.addressof(buf), len(data), C.byref(p), C.byref(w), C.byref(h)))
    return _take(p, w.value * h.value * 4, np.uint8, (h.value, w.value, 4))


def load_material_dir(renderer, path):
    m = C.c_int32()
    _check(_lib().bba_load_material_dir(renderer._ctx, str(path).encode(), C.byref(m)))
    return m.value


def load_material_set(renderer, pbr_root, capacity=64):
    """-> list of (material handle, directory name) in the reference's order (src/render.cpp:1243-1316)"""
    ids = (C.c_int32 * capacity)()
    names = ((C.c_char * 64) * capacity)()
    n = C.c_uint32()
    _check(_lib().bba_load_material_set(renderer._ctx, str(pbr_root).encode(), C.addressof(ids), C.addressof(names), capacity,
                                        C.byref(n)))
    return [(ids[i], names[i].value.decode()) for i in range(n.value)]
