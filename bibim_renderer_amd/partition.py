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
