"""ctypes binding of libbibim_hip.so (C ABI declared in include/bibim_hip.h).

There is no CPU fallback here or in the library: if the shared object is missing the import fails
loudly, and every compute entry point fails with BBR_ERR_NO_DEVICE when no HIP device is present.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BBR_LIB") or os.path.join(HERE, "libbibim_hip.so")  # (BBR_LIB: another build of the library, for A/B timing)

BBR_OK = 0
STATUS = {0: "BBR_OK", -1: "BBR_ERR_INVALID_ARGUMENT", -2: "BBR_ERR_NO_DEVICE", -3: "BBR_ERR_HIP",
          -4: "BBR_ERR_OUT_OF_MEMORY", -5: "BBR_ERR_BAD_HANDLE", -6: "BBR_ERR_NOT_IN_FRAME",
          -7: "BBR_ERR_TOO_MANY_PRIMITIVES", -8: "BBR_ERR_CAPACITY"}


class BbrImage(C.Structure):
    _fields_ = [("rgba", C.c_void_p), ("width", C.c_int32), ("height", C.c_int32)]


class BbrStats(C.Structure):
    _fields_ = [("n_prims", C.c_uint64), ("n_raster_tris", C.c_uint64), ("n_clipped_prims", C.c_uint64),
                ("n_bin_refs", C.c_uint64), ("n_broad_tris", C.c_uint64), ("n_shaded", C.c_uint64),
                ("bin_overflow", C.c_uint32), ("tile_w", C.c_uint32), ("tile_h", C.c_uint32), ("n_tiles", C.c_uint32)]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


class BibimError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"{STATUS.get(code, code)}: {msg}")
        self.code = code


# every symbol include/bibim_hip.h declares: name -> (restype, argtypes)
_P = C.c_void_p
SIGNATURES = {
    "bbr_create": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.POINTER(_P)]),
    "bbr_destroy": (C.c_int, [_P]),
    "bbr_resize": (C.c_int, [_P, C.c_int32, C.c_int32]),
    "bbr_stream_layout_state": (C.c_int, [_P, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_float)]),
    "bbr_last_error": (C.c_char_p, [_P]),
    "bbr_device_count": (C.c_int, []),
    "bbr_upload_mesh": (C.c_int, [_P, _P, C.c_uint32, _P, C.c_uint32, C.POINTER(C.c_int32)]),
    "bbr_upload_material": (C.c_int, [_P, C.POINTER(BbrImage), C.POINTER(C.c_int32)]),
    "bbr_free_mesh": (C.c_int, [_P, C.c_int32]),
    "bbr_free_material": (C.c_int, [_P, C.c_int32]),
    "bbr_set_frame_uniforms": (C.c_int, [_P, _P]),
    "bbr_set_view_uniforms": (C.c_int, [_P, _P]),
    "bbr_begin_frame": (C.c_int, [_P]),
    "bbr_draw": (C.c_int, [_P, C.c_int32, C.c_int32, _P, C.c_uint32]),
    "bbr_end_frame": (C.c_int, [_P]),
    "bbr_replay_frame": (C.c_int, [_P]),
    "bbr_synchronize": (C.c_int, [_P]),
    "bbr_read_framebuffer": (C.c_int, [_P, _P]),
    "bbr_framebuffer_device_ptr": (C.c_int, [_P, C.POINTER(_P), C.POINTER(C.c_uint64)]),
    "bbr_set_output_device_ptr": (C.c_int, [_P, _P, C.c_uint64]),
    "bbr_set_stream": (C.c_int, [_P, _P]),
    "bbr_set_partition": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32]),
    "bbr_shard_rows": (C.c_int, [_P, C.POINTER(C.c_int32)]),
    "bbr_read_shard": (C.c_int, [_P, _P]),
    "bbr_unpack_gathered": (C.c_int, [_P, _P, _P, _P]),
    "bbr_packed_shard_bytes": (C.c_int, [_P, C.POINTER(C.c_uint64)]),
    "bbr_pack_shard": (C.c_int, [_P, _P, _P]),
    "bbr_unpack_gathered_packed": (C.c_int, [_P, _P, _P, _P]),
    "bbr_wait_event": (C.c_int, [_P, _P]),
    "bbr_stream_wait_frame": (C.c_int, [_P, _P]),
    "bbr_tile_height": (C.c_int, [_P, C.POINTER(C.c_int32)]),
    "bbr_get_stats": (C.c_int, [_P, C.POINTER(BbrStats)]),
    "bbr_read_visibility": (C.c_int, [_P, _P, _P]),
    "bbr_last_frame_time_ms": (C.c_int, [_P, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "bbr_timing_reset": (C.c_int, [_P]),
    "bbr_timing_summary": (C.c_int, [_P, C.POINTER(C.c_uint32), C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float),
                                     C.POINTER(C.c_float)]),
    "bbr_set_option": (C.c_int, [_P, C.c_char_p, C.c_int64]),
    "bbr_tone_map": (C.c_int, [_P, C.c_int32, C.c_float]),
    "bbr_selftest_rcp": (C.c_int, [_P, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64)]),
    "bbr_read_gbuffer": (C.c_int, [_P, C.c_void_p]),
    "bbr_upload_gizmo": (C.c_int, [_P, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32]),
    "bbr_draw_overlays": (C.c_int, [_P, C.c_int32]),
    "bbr_present": (C.c_int, [_P, C.c_void_p, C.c_int32]),
    "bbr_read_presented": (C.c_int, [_P, C.c_void_p]),
    "bbr_present_buffer": (C.c_int, [_P, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int32, C.c_float, C.c_int32, C.c_void_p]),
    "bbr_present_timing": (C.c_int, [_P, C.POINTER(C.c_uint32), C.POINTER(C.c_float)]),
    "bbr_presented_device_ptr": (C.c_int, [_P, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]),
    "bbr_unpack_gathered_rgba8": (C.c_int, [_P, C.c_void_p, C.c_void_p, C.c_void_p]),
    "bbr_comm_unique_id": (C.c_int, [_P, _P]),
    "bbr_comm_init": (C.c_int, [_P, C.c_int32, C.c_int32, _P]),
    "bbr_comm_destroy": (C.c_int, [_P]),
    "bbr_comm_probe": (C.c_int, [_P]),
    "bbr_comm_count": (C.c_int, [_P, C.POINTER(C.c_int32)]),
    "bbr_stage_shard": (C.c_int, [_P, C.c_int32, _P, _P]),
    "bbr_exchange_block_bytes": (C.c_int, [_P, C.c_int32, C.POINTER(C.c_uint64)]),
    "bbr_allgather_frame": (C.c_int, [_P, C.c_int32, _P, _P, _P]),
    "bbr_push_shard": (C.c_int, [_P, C.c_int32, C.POINTER(_P), C.POINTER(C.c_int32), _P]),
    "bbr_push_state": (C.c_int, [_P, C.POINTER(C.c_int32)]),
    "bbr_capacity_growths": (C.c_int, [_P, C.POINTER(C.c_uint32)]),
    "bbr_host_timing": (C.c_int, [_P, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "bbr_host_timing_reset": (C.c_int, [_P]),
    "bbr_unpack_whole": (C.c_int, [_P, C.c_int32, _P, _P, _P]),
    "bbr_whole_frame_device_ptr": (C.c_int, [_P, C.POINTER(_P), C.POINTER(C.c_uint64)]),
    "bbr_read_whole_frame": (C.c_int, [_P, _P]),
    "bbr_device_alloc": (C.c_int, [_P, C.c_uint64, C.POINTER(_P)]),
    "bbr_device_free": (C.c_int, [_P, _P]),
    "bbr_copy_to_host": (C.c_int, [_P, _P, _P, C.c_uint64]),
    "bbr_ipc_export": (C.c_int, [_P, _P, _P]),
    "bbr_ipc_open": (C.c_int, [_P, _P, C.POINTER(_P)]),
    "bbr_ipc_close": (C.c_int, [_P, _P]),
}
COMM_ID_BYTES, IPC_HANDLE_BYTES = 128, 64
SHARD_RGBA32F, SHARD_PACKED, SHARD_RGBA8, SHARD_RGBA16F = 0, 1, 2, 3

# every symbol include/bibim_scene.h declares (C surface of the C++ Scene/Camera/drawFrame shim)
_F = C.c_float
SCENE_SIGNATURES = {
    "bbs_mat4_mul": (None, [_P, _P, _P]),
    "bbs_mat4_inverse": (None, [_P, _P]),
    "bbs_mat4_translate": (None, [_F, _F, _F, _P]),
    "bbs_mat4_scale": (None, [_F, _F, _F, _P]),
    "bbs_mat4_rotate": (None, [C.c_int, _F, _P]),
    "bbs_mat4_look_at": (None, [_P, _P, _P, _P]),
    "bbs_mat4_perspective": (None, [_F, _F, _F, _F, _P]),
    "bbs_camera_look": (None, [_F, _F, _P]),
    "bbs_camera_view": (None, [_P, _F, _F, _P]),
    "bbs_plane_mesh": (None, [_P, _P]),
    "bbs_shaderball_scene_create": (_P, [_P, _P, C.c_uint32, C.c_int32]),
    "bbs_shaderball_scene_create_from_file": (_P, [_P, C.c_char_p, C.c_int32]),
    "bbs_triangle_scene_create": (_P, [_P]),
    "bbs_scene_destroy": (None, [_P]),
    "bbs_scene_set_lights": (C.c_int, [_P, _P, C.c_uint32]),
    "bbs_scene_set_render_pass": (C.c_int, [_P, C.c_int32]),
    "bbs_scene_num_lights": (C.c_uint32, [_P]),
    "bbs_scene_get_lights": (C.c_int, [_P, _P]),
    "bbs_scene_instances": (C.c_int, [_P, C.c_int32, _P, C.c_uint32, C.POINTER(C.c_uint32)]),
    "bbs_fill_uniforms": (C.c_int, [_P, _P, _F, _F, C.c_int32, C.c_int32, _F, _F, _F, _F, C.c_int32, C.c_int32, _P, _P]),
    "bbs_draw_frame": (C.c_int, [_P, _P, _P, _F, _F, C.c_int32, C.c_int32, _F, _F, _F, _F, C.c_int32, C.c_int32, C.c_int32]),
}

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in list(SIGNATURES.items()) + list(SCENE_SIGNATURES.items()):
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib
