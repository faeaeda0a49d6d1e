"""Screen-band partition: index arithmetic, and the N > 1 data path on CPU with world_size 2
over gloo -- each rank renders only its bands (CPU oracle standing in for the GPU), packs its shard, the shards
are all-gathered and un-interleaved, and the result must equal the single-rank frame bit for bit."""
import os
import socket

import numpy as np
import pytest

from bibim_renderer_amd import partition as P


@pytest.mark.parametrize("H,world,band", [(2160, 8, 64), (2160, 2, 64), (1080, 4, 64), (1080, 3, 32), (270, 8, 32), (64, 8, 64), (100, 1, 32)])
def test_band_map_is_a_partition(H, world, band):
    seen = np.zeros(H, np.int32)
    for r in range(world):
        rows = P.owned_rows(H, r, world, band)
        seen[rows] += 1
        assert len(rows) <= P.shard_rows(H, world, band)
        rk, sr = P.shard_row_of(H, world, band)
        assert (rk[rows] == r).all() and np.array_equal(sr[rows], np.arange(len(rows)) if world > 1 else rows)
    assert (seen == 1).all()
    # interleaving: consecutive bands go to consecutive ranks
    if world > 1 and H > band:
        assert P.shard_row_of(H, world, band)[0][band] == 1 % world


def test_pack_unpack_round_trip():
    rng = np.random.Generator(np.random.PCG64(3))
    for H, world, band in ((270, 4, 32), (130, 3, 64), (64, 2, 32)):
        frame = rng.standard_normal((H, 17, 4)).astype(np.float32)
        gathered = np.stack([P.pack_shard(frame, r, world, band) for r in range(world)])
        assert gathered.shape[1] == P.shard_rows(H, world, band)
        assert np.array_equal(P.unpack_gathered(gathered, H, band), frame)


def test_packed_form_round_trip_and_layout():
    """rgb + one alpha bit per pixel: lossless for alpha in {0, 1} (what the path produces), NaN / inf colours included;
    block sizes: 12 bytes per pixel, masks 8-byte aligned, blocks 16-byte aligned"""
    rng = np.random.Generator(np.random.PCG64(5))
    for H, W, world, band in ((270, 17, 4, 32), (130, 390, 3, 64), (64, 64, 2, 32), (70, 1000, 8, 32)):
        frame = rng.standard_normal((H, W, 4)).astype(np.float32)
        frame[..., 3] = rng.integers(0, 2, (H, W)).astype(np.float32)
        frame[0, 0, :3] = (np.nan, np.inf, -0.0)
        rows = P.shard_rows(H, world, band)
        block, mask_offset = P.packed_layout(rows, W)
        assert block % 16 == 0 and mask_offset % 8 == 0 and mask_offset >= rows * W * 12
        assert block - mask_offset >= (rows * W + 7) // 8 and block < rows * W * 12.2 + 64
        gathered = np.concatenate([P.pack_shard_bits(P.pack_shard(frame, r, world, band)) for r in range(world)])
        assert gathered.size == world * block
        back = P.unpack_gathered_packed(gathered, H, W, world, band)
        assert np.array_equal(back.view(np.uint32), frame.view(np.uint32))


@pytest.mark.parametrize("world", [1, 2, 3, 4, 8])
def test_push_order_is_a_ring_step_at_every_step(world):
    """peer form: at step k the destinations of all ranks are a permutation (every rank receives exactly one block, every
    link direction carries one copy), and over the world - 1 steps every rank has sent to every other rank once"""
    orders = [P.push_order(r, world) for r in range(world)]
    for k in range(world - 1):
        assert sorted(o[k] for o in orders) == list(range(world))
    for r, o in enumerate(orders):
        assert sorted(o) == [x for x in range(world) if x != r]


@pytest.mark.parametrize("world", [1, 2, 3, 4, 8])
def test_direct_push_drives_every_link_of_the_mesh_at_once(world):
    """push_mode 1 (one kernel storing to all peers): a single step in which all world * (world - 1) link directions of the
    full mesh carry a block, each exactly once -- SURVEY 8(e)'s direct pattern; push_mode 0 needs world - 1 steps of
    `world` busy directions each, every rank sending and receiving on one link (ring timing)"""
    direct = P.push_links_busy(world, direct=True)
    every = {(a, b) for a in range(world) for b in range(world) if a != b}
    assert direct == ([every] if world > 1 else [])
    serial = P.push_links_busy(world, direct=False)
    assert len(serial) == max(world - 1, 0) and set().union(*serial) == every if world > 1 else serial == []
    for step in serial:
        assert len(step) == world and sorted(a for a, _ in step) == sorted(b for _, b in step) == list(range(world))
    # the same bytes land either way: destinations per rank are the same set
    for r in range(world):
        assert sorted(d for st in P.push_steps(r, world, True) for d in st) == sorted(d for st in P.push_steps(r, world, False) for d in st)


def test_binary16_wire_form_rounds_like_the_oracle():
    """BBR_SHARD_RGBA16F (the reference's HDR attachment format, src/render.h:94): the host model's conversion (numpy's
    float32 -> float16) is the oracle's bbo_half_round for every class of value -- every binary16 value, every midpoint
    between two of them and its neighbours, the overflow threshold, subnormals, infinities -- and a frame survives
    encode -> gather -> decode as its rounded self at 8 bytes per pixel"""
    from oracle import bbo
    h = np.arange(0, 0x7C00, dtype=np.uint16).view(np.float16).astype(np.float32)            # every finite binary16 value >= 0
    mid = ((h[:-1].astype(np.float64) + h[1:].astype(np.float64)) / 2).astype(np.float32)    # exact in binary32
    vals = np.concatenate([h, mid, np.nextafter(mid, np.float32(np.inf)), np.nextafter(mid, np.float32(-np.inf)),
                           np.float32([65504.0, 65519.99, 65520.0, 65536.0, 1e9, np.inf, 2.0 ** -25, 2.0 ** -24 * 1.5, 0.0])])
    vals = np.concatenate([vals, -vals]).astype(np.float32)
    with np.errstate(over="ignore"):
        via_numpy = vals.astype(np.float16).astype(np.float32)
    assert np.array_equal(via_numpy.view(np.uint32), bbo.half_round(vals).view(np.uint32))
    rng = np.random.Generator(np.random.PCG64(16))
    H, W, world, band = 130, 70, 3, 32
    frame = (rng.standard_normal((H, W, 4)) * np.float32(50)).astype(np.float32)
    frame[..., 3] = rng.integers(0, 2, (H, W)).astype(np.float32)
    block = P.exchange_block_bytes(P.SHARD_RGBA16F, H, W, world, band)
    assert block == P.shard_rows(H, world, band) * W * 8
    blocks = [P.encode_block(P.pack_shard(frame, r, world, band), P.SHARD_RGBA16F) for r in range(world)]
    assert all(b.size == block for b in blocks)
    back = P.decode_gathered(np.concatenate(blocks), P.SHARD_RGBA16F, H, W, world, band)
    assert back.dtype == np.float32 and np.array_equal(back.view(np.uint32), bbo.half_round(frame).view(np.uint32))
    assert np.array_equal(back[..., 3], frame[..., 3])                                        # alpha 0 / 1 is exact


@pytest.mark.parametrize("form", [P.SHARD_RGBA32F, P.SHARD_PACKED, P.SHARD_RGBA8])
@pytest.mark.parametrize("H,W,world,band", [(270, 17, 4, 32), (130, 390, 3, 64), (2160, 64, 8, 32), (64, 64, 1, 32)])
def test_peer_exchange_model_leaves_the_same_gather_buffer_on_every_rank(form, H, W, world, band):
    """every rank's gather buffer after all pushes == what ncclAllGather leaves there (blocks in rank order), and it
    decodes to the frame; block sizes are the ones bbr_exchange_block_bytes reports (16-byte multiples)"""
    rng = np.random.Generator(np.random.PCG64(H + W + world))
    if form == P.SHARD_RGBA8:
        frame = rng.integers(0, 256, (H, W, 4), dtype=np.uint8)
    else:
        frame = rng.standard_normal((H, W, 4)).astype(np.float32)
        frame[..., 3] = rng.integers(0, 2, (H, W)).astype(np.float32)
    block = P.exchange_block_bytes(form, H, W, world, band)
    assert block % 16 == 0 or form != P.SHARD_PACKED
    blocks = [P.encode_block(P.pack_shard(frame, r, world, band), form) for r in range(world)]
    assert all(b.size == block for b in blocks)
    bufs = [np.full(world * block, 0xAB, np.uint8) for _ in range(world)]
    for r in range(world):
        bufs[r][P.push_offset(r, block):P.push_offset(r, block) + block] = blocks[r]   # staged in the rank's own buffer
        for dst in P.push_order(r, world):
            bufs[dst][P.push_offset(r, block):P.push_offset(r, block) + block] = blocks[r]
    want = np.concatenate(blocks)
    for b in bufs:
        assert np.array_equal(b, want)
    back = P.decode_gathered(bufs[-1], form, H, W, world, band)
    assert back.dtype == frame.dtype and np.array_equal(back.view(np.uint8), frame.view(np.uint8))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _rank_main(rank, world, port, band_rows, q):
    import torch
    import torch.distributed as dist
    from bibim_renderer_amd import configs, textures
    from oracle import bbo, scenes
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = configs.C3.scaled(192, 150, 32)  # 150 rows: the last band is partial and the band count is odd
    sc = scenes.shaderball_scene(cfg, bbo.MaterialData(textures.make_material(32)))
    H, W = cfg.height, cfg.width
    mine = np.zeros((H, W, 4), np.float32)
    n = 0
    for b in range(rank, P.n_bands(H, band_rows), world):       # render ONLY the owned bands
        y0, y1 = b * band_rows, min((b + 1) * band_rows, H)
        part, _, _, st = bbo.render(sc, y0, y1, want_prim=False, want_depth=False)
        mine[y0:y1] = part[y0:y1]; n += st["n_shaded"]
    shard = torch.from_numpy(P.pack_shard(mine, rank, world, band_rows))
    gathered = torch.empty((world * shard.shape[0],) + tuple(shard.shape[1:]), dtype=torch.float32)
    dist.all_gather_into_tensor(gathered, shard)  # concatenation along dim 0 == [rank][shard row]
    frame = P.unpack_gathered(gathered.view((world,) + tuple(shard.shape)).numpy(), H, band_rows)
    # the same exchange in the packed form bench.py gathers by default (rgb + one alpha bit per pixel)
    packed = torch.from_numpy(P.pack_shard_bits(shard.numpy()))
    gathered_packed = torch.empty((world * packed.numel(),), dtype=torch.uint8)
    dist.all_gather_into_tensor(gathered_packed, packed)
    frame_packed = P.unpack_gathered_packed(gathered_packed.numpy(), H, W, world, band_rows)
    assert np.array_equal(frame_packed.view(np.uint32), frame.view(np.uint32))
    # the peer form of the native exchange (bbr_push_shard): every rank sends its block to every rank's gather buffer in
    # P.push_order -- here as point-to-point messages; step k of all ranks together is one ring step
    for form in (P.SHARD_RGBA32F, P.SHARD_PACKED):
        block = P.exchange_block_bytes(form, H, W, world, band_rows)
        mine_block = torch.from_numpy(P.encode_block(shard.numpy(), form).copy())
        assert mine_block.numel() == block
        buf = torch.zeros(world * block, dtype=torch.uint8)
        buf[P.push_offset(rank, block):P.push_offset(rank, block) + block] = mine_block
        for k, dst in enumerate(P.push_order(rank, world), start=1):
            src = (rank - k) % world                      # whose block lands here in the same step
            req = dist.isend(mine_block, dst)
            dist.recv(buf[P.push_offset(src, block):P.push_offset(src, block) + block], src)
            req.wait()
        pushed = P.decode_gathered(buf.numpy(), form, H, W, world, band_rows)
        assert np.array_equal(pushed.view(np.uint32), frame.view(np.uint32))
    total = torch.tensor([n]); dist.all_reduce(total)
    if rank == 0:
        full, _, _, st = bbo.render(sc, want_prim=False, want_depth=False)
        q.put((bool(np.array_equal(frame.view(np.uint32), full.view(np.uint32))), int(total.item()) == st["n_shaded"]))
    dist.destroy_process_group()


@pytest.mark.parametrize("band_rows", [32, 64])
def test_world2_gloo_allgather_reassembles_the_frame(band_rows):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_main, args=(r, 2, port, band_rows, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    same, count_ok = q.get(timeout=10)
    assert same and count_ok


def test_exchange_model_on_random_shapes():
    """hypothesis: for any frame shape, world size and band height the blocks pushed in P.push_order leave the all-gather
    layout on every rank and decode to the frame, in all three block forms"""
    from hypothesis import given, settings, strategies as st

    @settings(max_examples=60, deadline=None)
    @given(st.integers(1, 200), st.integers(1, 90), st.integers(1, 9), st.sampled_from([32, 64, 96]), st.sampled_from([0, 1, 2]),
           st.integers(0, 2 ** 31 - 1))
    def run(H, W, world, band, form, seed):
        rng = np.random.Generator(np.random.PCG64(seed))
        if form == P.SHARD_RGBA8:
            frame = rng.integers(0, 256, (H, W, 4), dtype=np.uint8)
        else:
            frame = rng.standard_normal((H, W, 4)).astype(np.float32)
            frame[..., 3] = rng.integers(0, 2, (H, W)).astype(np.float32)
        block = P.exchange_block_bytes(form, H, W, world, band)
        blocks = [P.encode_block(P.pack_shard(frame, r, world, band), form) for r in range(world)]
        assert all(b.size == block for b in blocks)
        bufs = [np.zeros(world * block, np.uint8) for _ in range(world)]
        for r in range(world):
            for dst in [r] + P.push_order(r, world):
                bufs[dst][P.push_offset(r, block):P.push_offset(r, block) + block] = blocks[r]
        for b in bufs[1:]:
            assert np.array_equal(b, bufs[0])
        back = P.decode_gathered(bufs[0], form, H, W, world, band)
        assert np.array_equal(back.view(np.uint8), frame.view(np.uint8))
        # every framebuffer row belongs to exactly one rank, at most shard_rows of them per rank
        rk, sr = P.shard_row_of(H, world, band)
        assert rk.max() < world and sr.max() < P.shard_rows(H, world, band)
        assert len({(int(a), int(b)) for a, b in zip(rk, sr)}) == H

    run()
