"""Screen-band partition of a frame over R ranks; host-side description of what
bbr_set_partition / bbr_unpack_gathered do on the device.  Pure index arithmetic, no rendering.

Band b = framebuffer rows [b*band_rows, (b+1)*band_rows) belongs to rank b % R (interleaved for load
balance: the balls cluster mid-screen).  A rank's shard is its bands stacked in order, padded to
shard_rows(H, R, band_rows) rows so that every rank contributes an equal-sized block to the all-gather.
"""
from __future__ import annotations

import numpy as np


def n_bands(height, band_rows):
    return (height + band_rows - 1) // band_rows


def shard_rows(height, world, band_rows):
    return ((n_bands(height, band_rows) + world - 1) // world) * band_rows if world > 1 else height


def owned_rows(height, rank, world, band_rows):
    """Framebuffer rows of `rank`, in shard order."""
    rows = []
    for b in range(rank, n_bands(height, band_rows), world):
        rows.extend(range(b * band_rows, min((b + 1) * band_rows, height)))
    return np.asarray(rows, dtype=np.int64)


def shard_row_of(height, world, band_rows):
    """For every framebuffer row y: (rank, row inside that rank's shard)."""
    y = np.arange(height)
    band = y // band_rows
    return band % world, (band // world) * band_rows + (y - band * band_rows)


def pack_shard(frame, rank, world, band_rows):
    """frame [H, W, C] -> this rank's padded shard [shard_rows, W, C] (padding rows are zero)."""
    H = frame.shape[0]
    out = np.zeros((shard_rows(H, world, band_rows),) + frame.shape[1:], frame.dtype)
    rk, sr = shard_row_of(H, world, band_rows)
    mine = rk == rank
    out[sr[mine]] = frame[mine]
    return out


def unpack_gathered(gathered, height, band_rows):
    """gathered [R, shard_rows, W, C] (all-gather output) -> row-major frame [H, W, C]."""
    world = gathered.shape[0]
    rk, sr = shard_row_of(height, world, band_rows)
    return gathered[rk, sr]


# ---- the packed form of a shard (bbr_pack_shard / bbr_unpack_gathered_packed): rgb + one alpha bit per pixel ----

def packed_layout(shard_rows_, width):
    """(block bytes, byte offset of the alpha masks) of one rank's packed block: rgb[n][3] float32, padding to 8 bytes,
    one little-endian 64-bit mask per 64 pixels (bit k of word w = pixel 64 w + k), padding to 16 bytes."""
    n = shard_rows_ * width
    mask_offset = (n * 12 + 7) & ~7
    return (mask_offset + ((n + 63) // 64) * 8 + 15) & ~15, mask_offset


def pack_shard_bits(shard):
    """shard [rows, W, 4] float32 with alpha in {0, 1} -> packed block (uint8).  Lossless for what the path produces:
    alpha is 1.0 on shaded pixels and 0.0 on cleared ones."""
    rows, width = shard.shape[:2]
    n = rows * width
    block, mask_offset = packed_layout(rows, width)
    out = np.zeros(block, np.uint8)
    flat = np.ascontiguousarray(shard, np.float32).reshape(n, 4)
    out[:n * 12] = np.ascontiguousarray(flat[:, :3]).view(np.uint8).reshape(-1)
    bits = np.zeros(((n + 63) // 64) * 64, np.uint8)
    bits[:n] = flat[:, 3].view(np.uint32) == 0x3F800000
    out[mask_offset:mask_offset + bits.size // 8] = np.packbits(bits, bitorder="little")
    return out


def unpack_gathered_packed(gathered, height, width, world, band_rows):
    """gathered: world packed blocks back to back (uint8) -> row-major frame [H, W, 4] float32."""
    rows = shard_rows(height, world, band_rows)
    n = rows * width
    block, mask_offset = packed_layout(rows, width)
    g = np.ascontiguousarray(gathered, np.uint8).reshape(world, block)
    shards = np.zeros((world, rows, width, 4), np.float32)
    for r in range(world):
        shards[r, ..., :3] = g[r, :n * 12].view(np.float32).reshape(rows, width, 3)
        bits = np.unpackbits(g[r, mask_offset:mask_offset + ((n + 63) // 64) * 8], bitorder="little")[:n]
        shards[r, ..., 3] = bits.reshape(rows, width).astype(np.float32)
    return unpack_gathered(shards, height, band_rows)


# ---- the native exchange (bbr_allgather_frame / bbr_push_shard, include/bibim_hip.h): block sizes and copy order ----

SHARD_RGBA32F, SHARD_PACKED, SHARD_RGBA8, SHARD_RGBA16F = 0, 1, 2, 3
# bytes per pixel of the forms with a fixed pixel size (the packed form: packed_layout)
_PIXEL_BYTES = {SHARD_RGBA32F: 16, SHARD_RGBA8: 4, SHARD_RGBA16F: 8}


def exchange_block_bytes(form, height, width, world, band_rows):
    """bytes one rank contributes to the exchange (bbr_exchange_block_bytes)"""
    rows = shard_rows(height, world, band_rows)
    if form == SHARD_PACKED:
        return packed_layout(rows, width)[0]
    return rows * width * _PIXEL_BYTES[form]


def push_order(rank, world):
    """peer form: the ranks this rank copies its block to, in order -- nearest first (rank + 1, rank + 2, ...), so that
    step k of every rank together is a permutation: each rank receives exactly one block per step, as in a ring step"""
    return [(rank + k) % world for k in range(1, world)]


def push_steps(rank, world, direct=True):
    """what bbr_push_shard queues, as steps that run one after the other, each a list of destinations served at the same
    time.  direct (option push_mode 1): ONE kernel stores the block to every peer -- a single step with world - 1
    destinations, one xGMI link each.  Otherwise world - 1 copies in push_order, one link at a time."""
    order = push_order(rank, world)
    return [order] if (direct and order) else [[d] for d in order]


def push_links_busy(world, direct=True):
    """per step: the set of (source, destination) link directions in use over all ranks (a rank drives one link per
    destination of its step).  direct: one step with all world * (world - 1) directions of the full mesh; copies: world - 1
    steps of `world` directions each, every rank sending and receiving on exactly one link."""
    steps = [push_steps(r, world, direct) for r in range(world)]
    n = max((len(s) for s in steps), default=0)
    return [{(r, d) for r in range(world) if k < len(steps[r]) for d in steps[r][k]} for k in range(n)]


def push_offset(rank, block_bytes):
    """byte offset of `rank`'s block inside every rank's gather buffer (the same layout ncclAllGather produces)"""
    return rank * block_bytes


def encode_block(shard, form):
    """a rank's shard [rows, W, 4] as the bytes that travel (uint8): float32 RGBA, the packed form, RGBA8, or binary16 RGBA"""
    if form == SHARD_PACKED:
        return pack_shard_bits(shard)
    if form == SHARD_RGBA16F:   # every channel to the nearest binary16 value, ties to even (numpy's conversion; the tests pin it to the CPU checker's rounding)
        with np.errstate(over="ignore"):
            return np.ascontiguousarray(shard, np.float32).astype(np.float16).view(np.uint8).reshape(-1)
    return np.ascontiguousarray(shard, np.float32 if form == SHARD_RGBA32F else np.uint8).view(np.uint8).reshape(-1)


def decode_gathered(gathered, form, height, width, world, band_rows):
    """world blocks back to back (uint8) -> the whole frame [H, W, 4] (what bbr_unpack_whole leaves in `whole`)"""
    g = np.ascontiguousarray(gathered, np.uint8)
    if form == SHARD_PACKED:
        return unpack_gathered_packed(g, height, width, world, band_rows)
    rows = shard_rows(height, world, band_rows)
    if form == SHARD_RGBA16F:   # widened back to RGBA32F (every binary16 value is a binary32 value)
        return unpack_gathered(g.view(np.float16).reshape(world, rows, width, 4), height, band_rows).astype(np.float32)
    dt = np.float32 if form == SHARD_RGBA32F else np.uint8
    return unpack_gathered(g.view(dt).reshape(world, rows, width, 4), height, band_rows)
