"""Thin object wrapper over the C ABI.  Names follow the reference's vocabulary: a mesh is a vertex
(+ index) buffer, a material is the six-map PBRMaterial, a frame is begin -> draw* -> end."""
from __future__ import annotations

import ctypes as C
import weakref

import numpy as np

from . import _capi
from ._capi import BbrImage, BbrStats, BibimError, lib

MAP_NAMES = ("albedo", "metallic", "roughness", "ao", "normal", "height")


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class Renderer:
    def __init__(self, width, height, device=0):
        self._L = lib()
        self._ctx = C.c_void_p()
        rc = self._L.bbr_create(width, height, device, C.byref(self._ctx))
        if rc != 0:
            raise BibimError(rc, (self._L.bbr_last_error(None) or b"").decode())
        self.width, self.height = int(width), int(height)
        self._scenes = weakref.WeakSet()  # host-shim scenes hold meshes of this context: they must go first

    # -- plumbing --
    def _check(self, rc):
        if rc != 0:
            raise BibimError(rc, (self._L.bbr_last_error(self._ctx) or b"").decode())

    def close(self):
        if self._ctx:
            for sc in list(self._scenes):
                sc.close()
            self._L.bbr_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def resize(self, width, height):
        """onWindowResize: new extent, same meshes / materials / options."""
        self._check(self._L.bbr_resize(self._ctx, width, height))
        self.width, self.height = int(width), int(height)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- resources --
    def upload_mesh(self, vertices, indices=None):
        v = np.ascontiguousarray(vertices)
        assert v.dtype.itemsize == 44 or (v.dtype == np.float32 and v.shape[-1] == 11)
        n = v.shape[0]
        idx = None if indices is None else np.ascontiguousarray(indices, np.uint32)
        out = C.c_int32()
        self._check(self._L.bbr_upload_mesh(self._ctx, _ptr(v), n, _ptr(idx), 0 if idx is None else len(idx), C.byref(out)))
        return out.value

    def upload_material(self, maps=None):
        maps = maps or {}
        arr = (BbrImage * 6)()
        keep = []
        for i, name in enumerate(MAP_NAMES):
            a = maps.get(name)
            if a is None:
                arr[i] = BbrImage(None, 0, 0)
            else:
                a = np.ascontiguousarray(a, np.uint8)
                assert a.ndim == 3 and a.shape[2] == 4
                keep.append(a)
                arr[i] = BbrImage(a.ctypes.data, a.shape[1], a.shape[0])
        out = C.c_int32()
        self._check(self._L.bbr_upload_material(self._ctx, arr, C.byref(out)))
        return out.value

    def free_mesh(self, mesh):
        self._check(self._L.bbr_free_mesh(self._ctx, mesh))

    def free_material(self, material):
        self._check(self._L.bbr_free_material(self._ctx, material))

    # -- frame --
    def set_frame_uniforms(self, block):
        b = np.ascontiguousarray(block)
        assert b.nbytes == 6432
        self._check(self._L.bbr_set_frame_uniforms(self._ctx, _ptr(b)))

    def set_view_uniforms(self, block):
        b = np.ascontiguousarray(block)
        assert b.nbytes == 144
        self._check(self._L.bbr_set_view_uniforms(self._ctx, _ptr(b)))

    def begin_frame(self):
        self._check(self._L.bbr_begin_frame(self._ctx))

    def draw(self, mesh, material, instances):
        inst = np.ascontiguousarray(instances)
        assert inst.nbytes % 128 == 0
        self._check(self._L.bbr_draw(self._ctx, mesh, material, _ptr(inst), inst.nbytes // 128))

    def end_frame(self):
        self._check(self._L.bbr_end_frame(self._ctx))

    def replay_frame(self):
        self._check(self._L.bbr_replay_frame(self._ctx))

    def synchronize(self):
        self._check(self._L.bbr_synchronize(self._ctx))

    # -- output --
    def read_framebuffer(self):
        out = np.empty((self.height, self.width, 4), np.float32)
        self._check(self._L.bbr_read_framebuffer(self._ctx, _ptr(out)))
        return out

    def read_visibility(self):
        prim = np.empty((self.height, self.width), np.uint32)
        depth = np.empty((self.height, self.width), np.float32)
        self._check(self._L.bbr_read_visibility(self._ctx, _ptr(prim), _ptr(depth)))
        return prim, depth

    def stats(self):
        s = BbrStats()
        self._check(self._L.bbr_get_stats(self._ctx, C.byref(s)))
        return s.as_dict()

    def set_option(self, name, value):
        self._check(self._L.bbr_set_option(self._ctx, name.encode(), int(value)))

    def last_frame_time_ms(self):
        a, b = C.c_float(), C.c_float()
        self._check(self._L.bbr_last_frame_time_ms(self._ctx, C.byref(a), C.byref(b)))
        return a.value, b.value

    def timing_reset(self):
        self._check(self._L.bbr_timing_reset(self._ctx))

    def timing_summary(self):
        """(frames, avg frame ms, avg geometry ms, avg raster ms, avg shade ms) since timing_reset()."""
        n, f, g, r, t = C.c_uint32(), C.c_float(), C.c_float(), C.c_float(), C.c_float()
        self._check(self._L.bbr_timing_summary(self._ctx, C.byref(n), C.byref(f), C.byref(g), C.byref(r), C.byref(t)))
        return n.value, f.value, g.value, r.value, t.value

    def selftest_rcp(self, lo_bits=0, hi_bits=0x7FFFFFFF):
        n = C.c_uint64()
        self._check(self._L.bbr_selftest_rcp(self._ctx, lo_bits, hi_bits, C.byref(n)))
        return n.value

    def tone_map(self, enable, exposure):
        self._check(self._L.bbr_tone_map(self._ctx, int(enable), float(exposure)))

    def read_gbuffer(self):
        """deferred path only: [h, w, 4 attachments, 4] float32 (binary16 values)"""
        out = np.empty((self.height, self.width, 4, 4), np.float32)
        self._check(self._L.bbr_read_gbuffer(self._ctx, _ptr(out)))
        return out

    # -- overlay subpass (light markers + corner gizmo over the presented image) --
    def upload_gizmo(self, vertices, indices=None):
        """vertices: float32 [n, 9] = pos, colour, normal (bb::GizmoVertex); indices: uint32 [m] or None"""
        v = np.ascontiguousarray(vertices, np.float32).reshape(-1, 9)
        i = None if indices is None else np.ascontiguousarray(indices, np.uint32)
        self._check(self._L.bbr_upload_gizmo(self._ctx, _ptr(v), v.shape[0], None if i is None else _ptr(i), 0 if i is None else i.size))

    def draw_overlays(self, gizmo_extent=100):
        self._check(self._L.bbr_draw_overlays(self._ctx, int(gizmo_extent)))

    # -- presentation (tone map + sRGB + UNORM8) --
    def present(self, rgba8_device_ptr=None, hdr16=True):
        """queue k_present for the last frame; EnableToneMapping / Exposure come from its FrameUniformBlock"""
        self._check(self._L.bbr_present(self._ctx, C.c_void_p(rgba8_device_ptr) if rgba8_device_ptr else None, int(bool(hdr16))))

    def present_buffer(self, rgba32f_ptr, rgba8_ptr, n_pixels, enable, exposure, hdr16=True, stream=None):
        self._check(self._L.bbr_present_buffer(self._ctx, C.c_void_p(rgba32f_ptr), C.c_void_p(rgba8_ptr), int(n_pixels),
                                               int(enable), float(exposure), int(bool(hdr16)),
                                               C.c_void_p(stream) if stream else None))

    def read_presented(self):
        rows = self.shard_rows()
        out = np.empty((rows, self.width, 4), np.uint8)
        self._check(self._L.bbr_read_presented(self._ctx, _ptr(out)))
        return out

    def present_timing(self):
        """(launches, average k_present ms) since timing_reset(), option "timing" on"""
        n, t = C.c_uint32(), C.c_float()
        self._check(self._L.bbr_present_timing(self._ctx, C.byref(n), C.byref(t)))
        return n.value, t.value

    def presented_device_ptr(self):
        p, n = C.c_void_p(), C.c_uint64()
        self._check(self._L.bbr_presented_device_ptr(self._ctx, C.byref(p), C.byref(n)))
        return p.value, n.value

    def unpack_gathered_rgba8(self, gathered_ptr, frame_ptr, stream=None):
        self._check(self._L.bbr_unpack_gathered_rgba8(self._ctx, C.c_void_p(gathered_ptr), C.c_void_p(frame_ptr),
                                                      C.c_void_p(stream) if stream else None))

    def stream_layout_state(self):
        """(layout in use, decided?, [ms of the timed spans per layout])"""
        lay, dec, ms = C.c_int32(), C.c_int32(), (C.c_float * 3)()
        self._check(self._L.bbr_stream_layout_state(self._ctx, C.byref(lay), C.byref(dec), ms))
        return lay.value, bool(dec.value), [float(x) for x in ms]

    # -- multi-GPU partition --
    def set_partition(self, rank, world, band_rows=0):
        self._check(self._L.bbr_set_partition(self._ctx, rank, world, band_rows))

    def shard_rows(self):
        r = C.c_int32()
        self._check(self._L.bbr_shard_rows(self._ctx, C.byref(r)))
        return r.value

    def tile_height(self):
        r = C.c_int32()
        self._check(self._L.bbr_tile_height(self._ctx, C.byref(r)))
        return r.value

    def read_shard(self):
        out = np.empty((self.shard_rows(), self.width, 4), np.float32)
        self._check(self._L.bbr_read_shard(self._ctx, _ptr(out)))
        return out

    def set_output_device_ptr(self, ptr, nbytes):
        self._check(self._L.bbr_set_output_device_ptr(self._ctx, C.c_void_p(ptr), nbytes))

    def set_stream(self, stream_handle):
        self._check(self._L.bbr_set_stream(self._ctx, C.c_void_p(stream_handle)))

    def framebuffer_device_ptr(self):
        p, n = C.c_void_p(), C.c_uint64()
        self._check(self._L.bbr_framebuffer_device_ptr(self._ctx, C.byref(p), C.byref(n)))
        return p.value, n.value

    def unpack_gathered(self, gathered_ptr, frame_ptr, stream_handle=None):
        self._check(self._L.bbr_unpack_gathered(self._ctx, C.c_void_p(gathered_ptr), C.c_void_p(frame_ptr),
                                                C.c_void_p(stream_handle) if stream_handle else None))

    def packed_shard_bytes(self):
        n = C.c_uint64()
        self._check(self._L.bbr_packed_shard_bytes(self._ctx, C.byref(n)))
        return n.value

    def pack_shard(self, packed_ptr, stream_handle=None):
        """the last frame's shard as rgb[n][3] float + one alpha bit per pixel (lossless, 12.1 B per pixel)"""
        self._check(self._L.bbr_pack_shard(self._ctx, C.c_void_p(packed_ptr), C.c_void_p(stream_handle) if stream_handle else None))

    def unpack_gathered_packed(self, gathered_ptr, frame_ptr, stream_handle=None):
        self._check(self._L.bbr_unpack_gathered_packed(self._ctx, C.c_void_p(gathered_ptr), C.c_void_p(frame_ptr),
                                                       C.c_void_p(stream_handle) if stream_handle else None))

    # -- native exchange (include/bibim_hip.h, "native exchange") --
    def comm_unique_id(self) -> bytes:
        buf = (C.c_uint8 * _capi.COMM_ID_BYTES)()
        self._check(self._L.bbr_comm_unique_id(self._ctx, buf))
        return bytes(buf)

    def comm_init(self, rank, world, unique_id: bytes):
        assert len(unique_id) == _capi.COMM_ID_BYTES
        buf = (C.c_uint8 * _capi.COMM_ID_BYTES).from_buffer_copy(unique_id)
        self._check(self._L.bbr_comm_init(self._ctx, rank, world, buf))

    def comm_destroy(self):
        self._check(self._L.bbr_comm_destroy(self._ctx))

    def comm_probe(self):
        """can this process load librccl?  Not collective: ranks vote on it before they enter comm_init together."""
        self._check(self._L.bbr_comm_probe(self._ctx))

    def comm_count(self):
        """ranks in the context's communicator, as RCCL reports them (ncclCommCount)"""
        n = C.c_int32()
        self._check(self._L.bbr_comm_count(self._ctx, C.byref(n)))
        return int(n.value)

    def stage_shard(self, form, block_ptr, stream_handle=None):
        """this rank's block of the last frame in `form` -> block_ptr (exchange_block_bytes(form) bytes)"""
        self._check(self._L.bbr_stage_shard(self._ctx, form, C.c_void_p(block_ptr), C.c_void_p(stream_handle) if stream_handle else None))

    def exchange_block_bytes(self, form):
        n = C.c_uint64()
        self._check(self._L.bbr_exchange_block_bytes(self._ctx, form, C.byref(n)))
        return n.value

    def allgather_frame(self, form, gathered_ptr=None, whole_ptr=None, stream_handle=None):
        self._check(self._L.bbr_allgather_frame(self._ctx, form, C.c_void_p(gathered_ptr) if gathered_ptr else None,
                                                C.c_void_p(whole_ptr) if whole_ptr else None,
                                                C.c_void_p(stream_handle) if stream_handle else None))

    def push_shard(self, form, peer_gathered_ptrs, peer_devices, stream_handle=None):
        n = len(peer_gathered_ptrs)
        ptrs = (C.c_void_p * n)(*[C.c_void_p(p) for p in peer_gathered_ptrs])
        devs = (C.c_int32 * n)(*peer_devices)
        self._check(self._L.bbr_push_shard(self._ctx, form, ptrs, devs, C.c_void_p(stream_handle) if stream_handle else None))

    def capacity_growths(self):
        """capacity growths since the context was created (host-side counter: no synchronisation)"""
        n = C.c_uint32()
        self._check(self._L.bbr_capacity_growths(self._ctx, C.byref(n)))
        return int(n.value)

    def host_timing(self):
        """host side of the frame loop since host_timing_reset (no GPU call): frames submitted, ns inside the submit calls, ns of
        that spent blocked on a frame slot still in flight, frames in which the host was blocked"""
        f, sub, blk, nb = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_uint64()
        self._check(self._L.bbr_host_timing(self._ctx, C.byref(f), C.byref(sub), C.byref(blk), C.byref(nb)))
        return {"frames": int(f.value), "submit_ns": int(sub.value), "blocked_ns": int(blk.value), "blocked_frames": int(nb.value)}

    def host_timing_reset(self):
        self._check(self._L.bbr_host_timing_reset(self._ctx))

    def push_was_direct(self):
        """True if the last push_shard stored through the one-kernel direct form (option push_mode 1 and every peer mapped)"""
        d = C.c_int32()
        self._check(self._L.bbr_push_state(self._ctx, C.byref(d)))
        return bool(d.value)

    def unpack_whole(self, form, gathered_ptr, whole_ptr=None, stream_handle=None):
        self._check(self._L.bbr_unpack_whole(self._ctx, form, C.c_void_p(gathered_ptr), C.c_void_p(whole_ptr) if whole_ptr else None,
                                             C.c_void_p(stream_handle) if stream_handle else None))

    def whole_frame_device_ptr(self):
        p, n = C.c_void_p(), C.c_uint64()
        self._check(self._L.bbr_whole_frame_device_ptr(self._ctx, C.byref(p), C.byref(n)))
        return p.value, n.value

    def read_whole_frame(self, form=0):
        out = np.empty((self.height, self.width, 4), np.uint8 if form == _capi.SHARD_RGBA8 else np.float32)
        self._check(self._L.bbr_read_whole_frame(self._ctx, _ptr(out)))
        return out

    def ipc_export(self, device_ptr) -> bytes:
        buf = (C.c_uint8 * _capi.IPC_HANDLE_BYTES)()
        self._check(self._L.bbr_ipc_export(self._ctx, C.c_void_p(device_ptr), buf))
        return bytes(buf)

    def ipc_open(self, handle: bytes):
        buf = (C.c_uint8 * _capi.IPC_HANDLE_BYTES).from_buffer_copy(handle)
        p = C.c_void_p()
        self._check(self._L.bbr_ipc_open(self._ctx, buf, C.byref(p)))
        return p.value

    def ipc_close(self, ptr):
        self._check(self._L.bbr_ipc_close(self._ctx, C.c_void_p(ptr)))

    def wait_event(self, event_handle):
        self._check(self._L.bbr_wait_event(self._ctx, C.c_void_p(event_handle)))

    def stream_wait_frame(self, stream_handle):
        self._check(self._L.bbr_stream_wait_frame(self._ctx, C.c_void_p(stream_handle)))

    # -- convenience: submit a scene description made of numpy inputs --
    def render_scene(self, scene, handles=None):
        """scene: object with .frame, .view and .draws, each draw having .vertices, .indices, .instances and
        .material.maps (dict name -> uint8 [h, w, 4]).  Returns handles so that repeated frames reuse the
        uploaded meshes / materials."""
        if handles is None:
            handles = {"mesh": {}, "mat": {}}
        self.set_frame_uniforms(scene.frame)
        self.set_view_uniforms(scene.view)
        self.begin_frame()
        for d in scene.draws:
            mk = id(d.vertices)
            if mk not in handles["mesh"]:
                handles["mesh"][mk] = self.upload_mesh(d.vertices, d.indices)
            tk = id(d.material)
            if tk not in handles["mat"]:
                handles["mat"][tk] = self.upload_material(d.material.maps)
            self.draw(handles["mesh"][mk], handles["mat"][tk], d.instances)
        self.end_frame()
        return handles
