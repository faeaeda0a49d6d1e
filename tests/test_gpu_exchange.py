"""The native exchange of include/bibim_hip.h on the one GPU a test box has: the collective form with a one-rank RCCL
communicator (librccl opened by the library, ncclCommInitRank, ncclAllGather in place, un-interleave), the peer form
among several contexts that share the device (hipMemcpyPeerAsync degenerates to a device copy), and the IPC handles of
the multi-process peer form between two processes.  More than one GPU has not been run: DESIGN.md section 6."""
import os
import subprocess
import sys

import numpy as np
import pytest

from bibim_renderer_amd import Renderer, BibimError, configs, partition as P
from oracle import bbo, scenes

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def scene_and_frames(maps64):
    cfg = configs.C3.scaled(512, 300, 64)
    sc = scenes.shaderball_scene(cfg, bbo.MaterialData(maps64))
    sc.frame["enable_tone_mapping"], sc.frame["exposure"] = 1, 1.2
    ref, _, _, _ = bbo.render(sc)
    return cfg, sc, ref, bbo.present(ref, 1, 1.2)


def test_collective_form_with_a_one_rank_communicator(scene_and_frames):
    import torch
    cfg, sc, ref, ref8 = scene_and_frames
    r = Renderer(cfg.width, cfg.height)
    with pytest.raises(BibimError):
        r.allgather_frame(P.SHARD_RGBA32F)                 # nothing rendered
    with pytest.raises(BibimError):
        r.stage_shard(P.SHARD_RGBA16F, 1)                  # nothing rendered either
    with pytest.raises(BibimError):
        r.comm_count()                                     # no communicator yet
    r.comm_probe()                                         # librccl loads: NOT collective, safe on one rank alone
    r.set_partition(0, 1, 32)
    h = r.render_scene(sc)
    r.synchronize()                                        # first frame of a scene: sizes the capacities (re-renders)
    with pytest.raises(BibimError):
        r.allgather_frame(P.SHARD_RGBA32F)                 # no communicator
    uid = r.comm_unique_id()
    assert len(uid) == 128 and any(uid)
    r.comm_init(0, 1, uid)
    with pytest.raises(BibimError):
        r.comm_init(0, 1, uid)                             # already has one
    assert r.comm_count() == 1                             # what RCCL itself says about the communicator
    ref16 = bbo.half_round(ref)                            # the binary16 wire form: the oracle's rounding of every channel
    for form, want in ((P.SHARD_RGBA32F, ref), (P.SHARD_PACKED, ref), (P.SHARD_RGBA16F, ref16)):
        assert r.exchange_block_bytes(form) == P.exchange_block_bytes(form, cfg.height, cfg.width, 1, 32)
        h = r.render_scene(sc, h)
        r.allgather_frame(form)                            # library-owned gather buffer and whole frame
        got = r.read_whole_frame(form)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), form
    with pytest.raises(BibimError):
        r.allgather_frame(P.SHARD_RGBA8)                   # needs bbr_present
    h = r.render_scene(sc, h)
    r.present()
    r.allgather_frame(P.SHARD_RGBA8)
    assert np.array_equal(r.read_whole_frame(P.SHARD_RGBA8), ref8)
    # caller's buffers on the caller's stream, several frames queued back to back without a synchronising call
    st = torch.cuda.Stream()
    block = r.exchange_block_bytes(P.SHARD_PACKED)
    gathered = [torch.zeros(block, dtype=torch.uint8, device="cuda") for _ in range(2)]
    whole = [torch.zeros((cfg.height, cfg.width, 4), dtype=torch.float32, device="cuda") for _ in range(2)]
    torch.cuda.synchronize()
    for n in range(6):
        h = r.render_scene(sc, h)
        r.allgather_frame(P.SHARD_PACKED, gathered[n & 1].data_ptr(), whole[n & 1].data_ptr(), st.cuda_stream)
    st.synchronize()
    for w in whole:
        assert np.array_equal(w.cpu().numpy().view(np.uint32), ref.view(np.uint32))
    r.comm_destroy()
    r.comm_destroy()                                       # idempotent
    r.close()


@pytest.mark.parametrize("world,form,push_mode", [(2, P.SHARD_PACKED, 1), (3, P.SHARD_RGBA32F, 1), (4, P.SHARD_RGBA8, 1),
                                                   (8, P.SHARD_PACKED, 1), (8, P.SHARD_RGBA32F, 0), (3, P.SHARD_RGBA8, 0),
                                                   (2, P.SHARD_RGBA16F, 1), (4, P.SHARD_RGBA16F, 0), (8, P.SHARD_RGBA16F, 1)])
def test_peer_form_among_contexts_on_one_device(scene_and_frames, world, form, push_mode):
    """every rank pushes its block into every rank's gather buffer -- with ONE kernel storing to all peers (push_mode 1, the
    default) or with copies queued one behind the other (0); afterwards all buffers hold the same bytes (the layout
    ncclAllGather leaves), the host model of partition.py predicts them, and every rank unpacks the whole frame"""
    import torch
    cfg, sc, ref, ref8 = scene_and_frames
    band = 32
    rs = [Renderer(cfg.width, cfg.height) for _ in range(world)]
    for rank, r in enumerate(rs):
        r.set_option("push_mode", push_mode)
        r.set_partition(rank, world, band)
    block = rs[0].exchange_block_bytes(form)
    assert block == P.exchange_block_bytes(form, cfg.height, cfg.width, world, band)
    bufs = [torch.full((world * block,), 0xAB, dtype=torch.uint8, device="cuda") for _ in range(world)]
    torch.cuda.synchronize()
    ptrs, devs = [b.data_ptr() for b in bufs], [0] * world
    for r in rs:
        r.render_scene(sc)
        r.synchronize()          # (first frame: capacities)
        r.render_scene(sc)
        if form == P.SHARD_RGBA8:
            r.present()
        r.push_shard(form, ptrs, devs)
        assert r.push_was_direct() == bool(push_mode)
    for r in rs:
        r.synchronize()          # "all pushes have landed": the host's part of the peer form
    host = [b.cpu().numpy() for b in bufs]
    for hb in host[1:]:
        assert np.array_equal(hb, host[0])
    want = ref8 if form == P.SHARD_RGBA8 else (bbo.half_round(ref) if form == P.SHARD_RGBA16F else ref)
    assert np.array_equal(P.decode_gathered(host[0], form, cfg.height, cfg.width, world, band).view(np.uint8), want.view(np.uint8))
    for r, b in zip(rs, bufs):
        r.unpack_whole(form, b.data_ptr())
        assert np.array_equal(r.read_whole_frame(form).view(np.uint8), want.view(np.uint8))
        r.close()


_IPC_CHILD = r"""
import sys, numpy as np, torch
sys.path.insert(0, sys.argv[1])
from bibim_renderer_amd import Renderer
r = Renderer(64, 64)
handle = bytes.fromhex(sys.stdin.readline().strip())
p = r.ipc_open(handle)
# write a pattern into the parent's buffer through the mapping
src = torch.arange(4096, dtype=torch.int32, device="cuda")
import ctypes
hip = ctypes.CDLL("libamdhip64.so")
torch.cuda.synchronize()
assert hip.hipMemcpy(ctypes.c_void_p(p), ctypes.c_void_p(src.data_ptr()), 4096 * 4, 3) == 0
assert hip.hipDeviceSynchronize() == 0
r.ipc_close(p)
r.close()
print("done", flush=True)
"""


def test_ipc_handles_carry_a_gather_buffer_to_another_process():
    """multi-process peer form: a rank exports its gather buffer (bbr_ipc_export), a peer process opens it (bbr_ipc_open)
    and writes its block there.  The buffer is a plain hipMalloc allocation (torch tensors live inside a caching
    allocator's segments; whether their handles open is the allocator's business, not the library's)."""
    import ctypes
    import torch
    hip = ctypes.CDLL("libamdhip64.so")
    r = Renderer(64, 64)
    buf = ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(buf), 4096 * 4) == 0
    assert hip.hipMemset(buf, 0, 4096 * 4) == 0 and hip.hipDeviceSynchronize() == 0
    handle = r.ipc_export(buf.value)
    assert len(handle) == 64
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    child = subprocess.run([sys.executable, "-c", _IPC_CHILD, ROOT], input=handle.hex() + "\n", capture_output=True, text=True,
                           timeout=300, env=env)
    assert child.returncode == 0 and "done" in child.stdout, child.stderr[-2000:]
    out = np.zeros(4096, np.int32)
    assert hip.hipMemcpy(out.ctypes.data_as(ctypes.c_void_p), buf, 4096 * 4, 2) == 0
    assert np.array_equal(out, np.arange(4096, dtype=np.int32))
    assert hip.hipFree(buf) == 0
    r.close()
