"""GPU: the C++ Scene/Camera/drawFrame shim end to end, the screen-band partition on one device, and
size-independent properties at BASELINE's full sizes."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, assert_frame_close
from bibim_renderer_amd import Renderer, configs, textures
from bibim_renderer_amd import partition as P
from bibim_renderer_amd import scene as S
from oracle import bbo, scenes

pytestmark = pytest.mark.gpu


def test_draw_frame_through_the_shim_matches_oracle(maps256):
    cfg = configs.C3.scaled(640, 360, 256)
    r = Renderer(cfg.width, cfg.height)
    material = r.upload_material(maps256)
    scene, cam, settings = S.config_scene(r, cfg)
    S.draw_frame(r, scene, cam, settings, material)
    img = r.read_framebuffer()
    ref, rprim, _, _ = bbo.render(scenes.shaderball_scene(cfg, bbo.MaterialData(maps256)))
    prim, _ = r.read_visibility()
    assert np.array_equal(prim, rprim)
    assert_frame_close(img, ref)
    scene.close(); r.close()


def test_frame_call_submits_the_frame_draw_frame_submits(maps256):
    """S.frame_call (bench.py's step: draw_frame with its arguments marshalled once) renders the same frame, bit for bit, and keeps
    doing so over many submissions with frames in flight"""
    cfg = configs.C3.scaled(640, 360, 256)
    r = Renderer(cfg.width, cfg.height)
    r.set_option("frames_in_flight", 3)
    material = r.upload_material(maps256)
    scene, cam, settings = S.config_scene(r, cfg)
    S.draw_frame(r, scene, cam, settings, material)
    want = r.read_framebuffer()
    submit = S.frame_call(r, scene, cam, settings, material)
    for _ in range(50):
        submit()
    got = r.read_framebuffer()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    ref, _, _, _ = bbo.render(scenes.shaderball_scene(cfg, bbo.MaterialData(maps256)))
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    scene.close(); r.close()


def test_scene_render_pass_type_selects_the_deferred_path(maps256):
    """SceneBase::SceneRenderPassType (src/scene.h:77) drives drawFrame as it drives recordCommand (src/main.cpp:89-112)"""
    cfg = configs.C3.scaled(480, 270, 256)
    r = Renderer(cfg.width, cfg.height)
    material = r.upload_material(maps256)
    scene, cam, settings = S.config_scene(r, cfg)
    sc = scenes.shaderball_scene(cfg, bbo.MaterialData(maps256))
    fwd, _, _, _ = bbo.render(sc)
    dfr, _, _, _, _ = bbo.render_deferred(sc)
    S.draw_frame(r, scene, cam, settings, material)
    assert np.array_equal(r.read_framebuffer().view(np.uint32), fwd.view(np.uint32))
    scene.set_render_pass(True)
    S.draw_frame(r, scene, cam, settings, material)
    assert np.array_equal(r.read_framebuffer().view(np.uint32), dfr.view(np.uint32))
    scene.set_render_pass(False)
    S.draw_frame(r, scene, cam, settings, material)
    assert np.array_equal(r.read_framebuffer().view(np.uint32), fwd.view(np.uint32))
    scene.close(); r.close()


def test_reference_default_scene_through_the_shim(maps64):
    """ShaderBallScene as the reference constructs it: 1 instance, its 3 lights, normal map off, 1280x720 window"""
    r = Renderer(1280, 720)
    material = r.upload_material(maps64)
    scene = S.ShaderBallScene(r, grid=1)
    S.draw_frame(r, scene, S.FreeLookCamera(), S.FrameSettings(), material)
    img = r.read_framebuffer()
    osc = scenes.shaderball_scene(configs.C2.scaled(1280, 720, 64), bbo.MaterialData(maps64))
    osc.frame = scenes.frame_uniforms(scenes.reference_default_lights()); osc.view["enable_normal_map"] = 0
    ref, _, _, _ = bbo.render(osc)
    assert_frame_close(img, ref)
    scene.close(); r.close()


def test_triangle_scene_through_the_shim():
    r = Renderer(64, 64)
    material = r.upload_material({})
    scene = S.TriangleScene(r)
    S.draw_frame(r, scene, S.FreeLookCamera(), S.FrameSettings(), material)
    g = np.load(os.path.join(GOLDEN, "oracle_frames.npz"))
    assert_frame_close(r.read_framebuffer(), g["triangle64_rgba_bits"].view(np.float32))
    scene.close(); r.close()


@pytest.mark.parametrize("world,band_rows,tile_mode", [(2, 64, 0), (3, 64, 0), (8, 64, 0), (4, 32, 1), (8, 128, 0), (5, 96, 1)])
def test_partitioned_render_reassembles_to_the_single_gpu_frame(maps64, world, band_rows, tile_mode, item_route_heavy):
    """every rank's shard rendered on this one GPU in turn; host-side all-gather + un-interleave == full frame"""
    cfg = configs.C3.scaled(512, 300, 64)
    sc = scenes.shaderball_scene(cfg, bbo.MaterialData(maps64))
    ref, _, _, rst = bbo.render(sc)
    shards, n = [], 0
    for rank in range(world):
        r = Renderer(cfg.width, cfg.height)
        r.set_option("tile_mode", tile_mode)
        r.set_partition(rank, world, band_rows)
        assert r.shard_rows() == P.shard_rows(cfg.height, world, band_rows)
        r.render_scene(sc)
        shards.append(r.read_shard())
        n += r.stats()["n_shaded"]
        r.close()
    frame = P.unpack_gathered(np.stack(shards), cfg.height, band_rows)
    assert n == rst["n_shaded"]
    assert_frame_close(frame, ref)
    assert np.array_equal(frame.view(np.uint32), ref.view(np.uint32))


def test_presented_shards_reassemble(maps64):
    """RGBA8 shards (a quarter of the fp32 payload) gathered and un-interleaved on the device == presented full frame"""
    import torch
    cfg = configs.C3.scaled(512, 300, 64)
    sc = scenes.shaderball_scene(cfg, bbo.MaterialData(maps64))
    sc.frame["enable_tone_mapping"], sc.frame["exposure"] = 1, 1.2
    ref, _, _, _ = bbo.render(sc)
    want = bbo.present(ref, 1, 1.2)
    world, band_rows = 3, 32
    shards = []
    for rank in range(world):
        r = Renderer(cfg.width, cfg.height)
        r.set_partition(rank, world, band_rows)
        r.render_scene(sc)
        r.present()
        shards.append(r.read_presented())
        assert shards[-1].shape == (r.shard_rows(), cfg.width, 4)
        if rank == world - 1:
            gathered = torch.from_numpy(np.stack(shards)).cuda()
            frame = torch.zeros((cfg.height, cfg.width, 4), dtype=torch.uint8, device="cuda")
            torch.cuda.synchronize()   # torch fills on its own stream; the library does not wait for that one
            r.unpack_gathered_rgba8(gathered.data_ptr(), frame.data_ptr())
            r.synchronize()
            torch.cuda.synchronize()
            assert np.array_equal(frame.cpu().numpy(), want)
        r.close()
    host = P.unpack_gathered(np.stack(shards), cfg.height, band_rows)
    assert np.array_equal(host, want)


def test_device_side_unpack_matches_host_unpack(maps64):
    import torch
    cfg = configs.C3.scaled(384, 200, 64)
    sc = scenes.shaderball_scene(cfg, bbo.MaterialData(maps64))
    world, band = 4, 64
    rs = [Renderer(cfg.width, cfg.height) for _ in range(world)]
    shard_rows = P.shard_rows(cfg.height, world, band)
    gathered = torch.zeros((world, shard_rows, cfg.width, 4), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()   # torch fills on its own stream; the library does not wait for that one
    for rank, r in enumerate(rs):
        r.set_partition(rank, world, band)
        r.set_output_device_ptr(gathered[rank].data_ptr(), gathered[rank].numel() * 4)  # render straight into the gather slot
        r.render_scene(sc)
        r.synchronize()
    frame = torch.empty((cfg.height, cfg.width, 4), dtype=torch.float32, device="cuda")
    rs[0].unpack_gathered(gathered.data_ptr(), frame.data_ptr())
    rs[0].synchronize()
    torch.cuda.synchronize()
    host = P.unpack_gathered(gathered.cpu().numpy(), cfg.height, band)
    assert np.array_equal(frame.cpu().numpy().view(np.uint32), host.view(np.uint32))
    ref, _, _, _ = bbo.render(sc)
    assert_frame_close(frame.cpu().numpy(), ref)
    for r in rs:
        r.close()


@pytest.fixture(scope="module")
def c4_whole_frame():
    """C4 = C3's frame (3840x2160, 16 balls, 4 lights, 2048^2 maps) rendered unpartitioned; that frame is compared with
    the oracle bit for bit in test_gpu_parity.py::test_c3_full_size_4k"""
    cfg = configs.C4
    maps = textures.make_material(cfg.texture_size)
    sc = scenes.shaderball_scene(cfg, bbo.MaterialData(maps))
    r = Renderer(cfg.width, cfg.height)
    h = r.render_scene(sc)
    want = r.read_framebuffer()
    st = r.stats()
    r.close()
    assert st["n_shaded"] == json.load(open(os.path.join(GOLDEN, "n_shaded.json")))["c3"]["n_shaded"]
    return cfg, sc, want


@pytest.mark.parametrize("world", [2, 4, 8])
def test_c4_full_size_4k_split_over_2_4_8_ranks(c4_whole_frame, world):
    """BASELINE config #4 at full size: the 4K frame split into interleaved tile-high bands over `world` ranks, every
    rank's shard rendered on this one GPU, gathered in both payload forms bench.py offers -- rgba32f shards rendered
    straight into the gather slots, and packed shards (rgb + alpha bit) -- and unpacked on the device: the frame every rank
    ends up with is the unpartitioned frame, bit for bit, and the ranks' coverage counts add up to the oracle's."""
    import torch
    cfg, sc, want = c4_whole_frame
    rs = [Renderer(cfg.width, cfg.height) for _ in range(world)]
    band = rs[0].tile_height()
    shard_rows = P.shard_rows(cfg.height, world, band)
    gathered = torch.zeros((world, shard_rows, cfg.width, 4), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    handles, n_shaded, pb = None, 0, None
    for rank, r in enumerate(rs):
        r.set_partition(rank, world, band)
        assert r.shard_rows() == shard_rows
        pb = r.packed_shard_bytes()
        r.set_output_device_ptr(gathered[rank].data_ptr(), gathered[rank].numel() * 4)
        r.render_scene(sc)
        r.synchronize()
        n_shaded += r.stats()["n_shaded"]
    assert n_shaded == json.load(open(os.path.join(GOLDEN, "n_shaded.json")))["c3"]["n_shaded"]
    packed = torch.full((world * pb,), 0xAB, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    for rank, r in enumerate(rs):
        r.pack_shard(packed[rank * pb:].data_ptr())
        r.synchronize()
    # ... and in the reference's own HDR attachment format as a wire form (BBR_SHARD_RGBA16F, 8 bytes per pixel, lossy):
    # every channel of the frame rounded to binary16 exactly as the oracle rounds it
    hb = rs[0].exchange_block_bytes(P.SHARD_RGBA16F)
    assert hb == shard_rows * cfg.width * 8
    halves = torch.full((world * hb,), 0xAB, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    for rank, r in enumerate(rs):
        r.stage_shard(P.SHARD_RGBA16F, halves[rank * hb:].data_ptr())
        r.synchronize()
    for form in ("rgba32f", "packed", "rgba16f"):
        frame = torch.full((cfg.height, cfg.width, 4), 3.0, dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        if form == "rgba32f":
            rs[-1].unpack_gathered(gathered.data_ptr(), frame.data_ptr())
        elif form == "packed":
            rs[-1].unpack_gathered_packed(packed.data_ptr(), frame.data_ptr())
        else:
            rs[-1].unpack_whole(P.SHARD_RGBA16F, halves.data_ptr(), frame.data_ptr())
        rs[-1].synchronize()
        torch.cuda.synchronize()
        got = frame.cpu().numpy()
        expect = bbo.half_round(want) if form == "rgba16f" else want
        assert np.array_equal(got.view(np.uint32), expect.view(np.uint32)), form
    for r in rs:
        r.close()


def test_c5_8k_properties():
    """BASELINE config #5 at full size (7680x4320, 64 balls, 8 lights, 2048^2 maps): coverage count against the oracle's
    committed N_shaded, determinism, and the whole frame against the oracle, bit for bit."""
    cfg = configs.C5
    maps = textures.make_material(cfg.texture_size)   # 2048^2 maps, as BASELINE's configuration has them
    cfgs = cfg
    r = Renderer(cfg.width, cfg.height)
    material = r.upload_material(maps)
    scene, cam, settings = S.config_scene(r, cfgs)
    S.draw_frame(r, scene, cam, settings, material)
    a = r.read_framebuffer()
    st = r.stats()
    want = json.load(open(os.path.join(GOLDEN, "n_shaded.json")))["c5"]
    assert st["n_shaded"] == want["n_shaded"] and st["n_prims"] == want["n_prims"]
    r.replay_frame()
    assert np.array_equal(r.read_framebuffer().view(np.uint32), a.view(np.uint32))
    # the whole 8K frame against the oracle, bit for bit: bands of 48 rows over the host's cores (ctypes drops the GIL)
    from concurrent.futures import ThreadPoolExecutor
    osc = scenes.shaderball_scene(cfgs, bbo.MaterialData(maps))

    def band(y0):
        ref, _, _, _ = bbo.render(osc, y0, min(y0 + 48, cfg.height), want_prim=False, want_depth=False)
        return y0, bool(np.array_equal(ref[y0:y0 + 48].view(np.uint32), a[y0:y0 + 48].view(np.uint32)))

    with ThreadPoolExecutor(min(16, os.cpu_count() or 1)) as ex:
        wrong = [y0 for y0, ok in ex.map(band, range(0, cfg.height, 48)) if not ok]
    assert not wrong, wrong[:8]
    # alpha is 1 exactly on geometry and 0 on background; background colour is the clear colour
    cov = a[..., 3] == 1.0
    assert int(cov.sum()) == want["n_shaded"] and (a[~cov] == 0).all()
    scene.close(); r.close()


def test_light_superposition_property_at_full_hd(maps256):
    """size-independent property: with ambient removed, radiance from lights {A,B} = radiance(A) + radiance(B)
    up to float summation error (Lo is a sum over lights in the shader)."""
    cfg = configs.C2.scaled(1920, 1080, 256)
    base = scenes.shaderball_scene(cfg, bbo.MaterialData(maps256))
    la = scenes.light(0, pos=(0, 2, 0), color=(1, 0.8, 0.8), intensity=50.0)
    lb = scenes.light(0, pos=(2, 1, 1), color=(0.8, 1, 0.8), intensity=20.0)
    r = Renderer(cfg.width, cfg.height)
    outs = []
    h = None
    for lights in ([], [la], [lb], [la, lb]):
        base.frame = scenes.frame_uniforms(lights)
        h = r.render_scene(base, h)
        outs.append(r.read_framebuffer().astype(np.float64))
    r.close()
    amb, a, b, ab = outs
    np.testing.assert_allclose((a - amb) + (b - amb), ab - amb, rtol=2e-5, atol=2e-5)


def test_material_directories_load_like_create_pbr_material_set(maps64, tmp_path):
    """pbr/<name>/{albedo,metallic,roughness,ao,normal,height}.png -> materials in the reference's order
    (src/render.cpp:1243-1316): directories by name, "default" swapped with the last and dropped, missing maps fall
    back to the default maps; the loaded material renders exactly like the same maps handed over directly"""
    from test_assets import encode_png
    from bibim_renderer_amd import assets

    def write_dir(name, maps):
        d = tmp_path / "pbr" / name
        d.mkdir(parents=True)
        for k, img in maps.items():
            a = np.asarray(img)
            if k in ("metallic", "roughness", "ao"):      # as most reference maps: grey files, expanded on load
                (d / f"{k}.png").write_bytes(encode_png(a[..., :1].astype(int), 0, 8))
            else:
                (d / f"{k}.png").write_bytes(encode_png(a[..., :3].astype(int), 2, 8))
    grey = {k: np.repeat(v[..., :1], 4, axis=2) for k, v in maps64.items() if k in ("metallic", "roughness", "ao")}
    for k in grey:
        grey[k][..., 3] = 255
    full = {**{k: np.concatenate([v[..., :3], np.full(v.shape[:2] + (1,), 255, np.uint8)], axis=2) for k, v in maps64.items()}, **grey}
    write_dir("zinc", {k: full[k] for k in ("albedo", "normal")})            # partial: the rest falls back to default
    write_dir("bark1", full)
    write_dir("default", {"albedo": np.full((4, 4, 4), 255, np.uint8)})
    write_dir("alder", {k: full[k] for k in ("roughness", "metallic")})
    (tmp_path / "pbr" / "notes.txt").write_text("not a directory")
    cfg = configs.C2.scaled(256, 144, 64)
    r = Renderer(cfg.width, cfg.height)
    loaded = assets.load_material_set(r, tmp_path / "pbr")
    assert [n for _, n in loaded] == ["alder", "bark1", "zinc"]             # [alder bark1 default zinc] -> default<->zinc, pop
    for (mid, name), maps in zip(loaded, ({k: full[k] for k in ("roughness", "metallic")}, full, {k: full[k] for k in ("albedo", "normal")})):
        sc = scenes.shaderball_scene(cfg, bbo.MaterialData(maps))
        ref, _, _, _ = bbo.render(sc)
        r.set_frame_uniforms(sc.frame); r.set_view_uniforms(sc.view)
        r.begin_frame()
        for d in sc.draws:
            r.draw(r.upload_mesh(d.vertices, d.indices), mid, d.instances)
        r.end_frame()
        assert np.array_equal(r.read_framebuffer().view(np.uint32), ref.view(np.uint32)), name
    single = assets.load_material_dir(r, tmp_path / "pbr" / "bark1")
    assert single not in [m for m, _ in loaded]
    with pytest.raises(assets.AssetError):
        assets.load_material_set(r, tmp_path / "nope")
    r.close()


def test_frames_stay_identical_while_other_processes_share_the_gpu(maps64):
    """Regression for two races that only showed with other processes on the GPU (waves of a workgroup starting far
    apart): k_raster reading a tile's bin counts after another wave had cleared them, and zero fills on the NULL stream
    landing after the first kernels.  Two contender processes render in a loop; this one renders 300 partitioned frames
    into two alternating caller buffers (bench.py's N > 1 pattern) and every frame must have the same bits.
    Third pass (round 4): the long frames' route with k_raster's heavy slots, where a tile's bin counts are read by two
    workgroups that may start far apart."""
    import subprocess, sys, time, torch
    contender = ("import sys; sys.path.insert(0, '.');\n"
                 "from bibim_renderer_amd import configs, textures, Renderer; from bibim_renderer_amd import scene as S\n"
                 "cfg = configs.C3.scaled(1920, 1080, 256); r = Renderer(cfg.width, cfg.height)\n"
                 "m = r.upload_material(textures.make_material(256)); sc, cam, st = S.config_scene(r, cfg)\n"
                 "import time; t = time.time()\n"
                 "while time.time() - t < 16: [S.draw_frame(r, sc, cam, st, m) for _ in range(50)]; r.synchronize()\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    procs = [subprocess.Popen([sys.executable, "-c", contender], cwd=root, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
             for _ in range(2)]
    try:
        time.sleep(4.0)  # let the contenders get going (their first import of torch-free modules is quick)
        cfg = configs.C3.scaled(1920, 1080, 64)
        for fif, heavy in ((1, 0), (3, 0), (3, 4)):
            r = Renderer(cfg.width, cfg.height)
            r.set_option("frames_in_flight", fif)
            if heavy:
                r.set_option("no_tail_items", 0)
                r.set_option("heavy_tiles", heavy)
            material = r.upload_material(maps64)
            scene, cam, settings = S.config_scene(r, cfg)
            r.set_partition(1, 4, r.tile_height())
            S.draw_frame(r, scene, cam, settings, material)
            r.synchronize()          # first frame sizes the bins (an overflowed frame is re-rendered here)
            rows, W, steps = r.shard_rows(), cfg.width, 300
            # zeros: the rows of a band that hang over the bottom of the frame are padding nobody writes
            shard = [torch.zeros((rows, W, 4), dtype=torch.float32, device="cuda") for _ in range(2)]
            keep = torch.empty((steps, rows, W, 4), dtype=torch.float32, device="cuda")
            torch.cuda.synchronize()   # torch fills on its own stream; the library does not wait for that one
            side = torch.cuda.Stream()
            consumed = [torch.cuda.Event(), torch.cuda.Event()]
            for n in range(steps):
                b = n & 1
                if n >= 2:
                    r.wait_event(consumed[b].cuda_event)
                r.set_output_device_ptr(shard[b].data_ptr(), shard[b].numel() * 4)
                S.draw_frame(r, scene, cam, settings, material)
                r.stream_wait_frame(side.cuda_stream)
                with torch.cuda.stream(side):
                    keep[n].copy_(shard[b], non_blocking=True)
                    consumed[b].record(side)
            r.synchronize()
            torch.cuda.synchronize()
            k = keep.view(torch.int32)
            bad = [n for n in range(steps) if not torch.equal(k[n], k[0])]
            assert not bad, (fif, heavy, bad[:10])
            scene.close(); r.close()
    finally:
        for p in procs:
            p.wait(timeout=60)


def test_a_host_that_never_synchronises_still_outgrows_an_overflow(maps64, item_route_heavy):
    """bins far too small, frames only streamed into a caller buffer: the first frames are incomplete (and stay so),
    but within a few frames the capacities have grown by themselves and every later frame is exact"""
    import torch
    cfg = configs.C3.scaled(640, 360, 64)
    sc = scenes.shaderball_scene(cfg, bbo.MaterialData(maps64))
    ref, _, _, _ = bbo.render(sc)
    for fif in (1, 2, 3):
        r = Renderer(cfg.width, cfg.height)
        r.set_option("frames_in_flight", fif)
        r.set_option("bin_cap", 4)
        steps = 16
        out = torch.zeros((cfg.height, cfg.width, 4), dtype=torch.float32, device="cuda")
        keep = torch.empty((steps,) + tuple(out.shape), dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()   # torch fills on its own stream; the library does not wait for that one
        side = torch.cuda.Stream()
        r.set_output_device_ptr(out.data_ptr(), out.numel() * 4)
        h = None
        done = torch.cuda.Event()
        for n in range(steps):
            if n:
                r.wait_event(done.cuda_event)
            h = r.render_scene(sc, h)                 # asynchronous; nothing here synchronises the context
            r.stream_wait_frame(side.cuda_stream)
            with torch.cuda.stream(side):
                keep[n].copy_(out, non_blocking=True)
                done.record(side)
        torch.cuda.synchronize()
        frames = keep.cpu().numpy()
        exact = [bool(np.array_equal(f.view(np.uint32), ref.view(np.uint32))) for f in frames]
        assert not exact[0]                           # 4 references per bin cannot hold this scene
        first_good = exact.index(True)
        assert first_good <= 3 * (fif + 1), exact     # a growth step or two, each a few frames after the overflow
        assert all(exact[first_good:]), exact         # and it stays right
        assert r.stats()["bin_overflow"] >= 1
        r.close()


def test_two_and_a_half_million_triangles_at_4k(maps256):
    """256 ShaderBalls (2 502 658 triangles, most of them sub-pixel) on a 4K frame: the whole image against the oracle,
    bit for bit (band-parallel oracle), with the first-frame bin overflow on the way"""
    from concurrent.futures import ThreadPoolExecutor
    from dataclasses import replace
    cfg = replace(configs.C5, width=3840, height=2160, grid=16, cam_pos=(0.0, 8.0, -10.0), cam_pitch=-25.0, texture_size=256,
                  name="big")
    sc = scenes.shaderball_scene(cfg, bbo.MaterialData(maps256))
    r = Renderer(cfg.width, cfg.height)
    r.render_scene(sc)
    a = r.read_framebuffer()
    st = r.stats()
    assert st["n_prims"] == 256 * 9776 + 2 and st["n_raster_tris"] > 1_000_000

    def band(y0):
        ref, _, _, _ = bbo.render(sc, y0, min(y0 + 24, cfg.height), want_prim=False, want_depth=False)
        return y0, bool(np.array_equal(ref[y0:y0 + 24].view(np.uint32), a[y0:y0 + 24].view(np.uint32)))

    with ThreadPoolExecutor(min(16, os.cpu_count() or 1)) as ex:
        wrong = [y0 for y0, ok in ex.map(band, range(0, cfg.height, 24)) if not ok]
    assert not wrong, wrong[:8]
    r.close()


@pytest.mark.parametrize("width,height,world,band,deferred", [(384, 200, 4, 64, 0), (390, 203, 3, 32, 0), (384, 200, 2, 32, 1),
                                                              (1000, 70, 8, 32, 0)])
def test_packed_shards_reassemble_the_same_frame(maps64, width, height, world, band, deferred, item_route):
    """the all-gather payload as rgb + one alpha bit per pixel (bbr_pack_shard / bbr_unpack_gathered_packed): the frame
    every rank ends up with is the unpartitioned frame, bit for bit -- forward (alpha 0 on cleared pixels, 1 on shaded
    ones) and deferred (1 everywhere), widths off the 64-pixel mask grid, more ranks than bands"""
    import torch
    cfg = configs.C3.scaled(width, height, 64)
    sc = scenes.shaderball_scene(cfg, bbo.MaterialData(maps64))
    whole = Renderer(cfg.width, cfg.height)
    whole.set_option("render_pass", deferred)
    whole.render_scene(sc)
    want = whole.read_framebuffer()
    whole.close()
    assert set(np.unique(want[..., 3]).tolist()) <= {0.0, 1.0}
    rs = [Renderer(cfg.width, cfg.height) for _ in range(world)]
    pb = None
    for rank, r in enumerate(rs):
        r.set_option("render_pass", deferred)
        r.set_partition(rank, world, band)
        assert pb in (None, r.packed_shard_bytes())
        pb = r.packed_shard_bytes()
    shard_rows = rs[0].shard_rows()
    assert pb % 16 == 0 and pb >= shard_rows * cfg.width * 12 + (shard_rows * cfg.width + 7) // 8
    gathered = torch.full((world * pb,), 0xAB, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()                                # torch's fill runs on torch's stream, the library on its own
    for rank, r in enumerate(rs):
        r.render_scene(sc)
        r.synchronize()                                     # (a first frame may be re-rendered with larger capacities)
        r.render_scene(sc)
        r.pack_shard(gathered[rank * pb:].data_ptr())       # on the frame's own stream: straight into the gather slot
        r.synchronize()
    torch.cuda.synchronize()
    # the device's bytes are the host model's (bibim_renderer_amd/partition.py), padding aside
    g = gathered.cpu().numpy().reshape(world, pb)
    block, mask_offset = P.packed_layout(shard_rows, cfg.width)
    assert block == pb
    n = shard_rows * cfg.width
    for rank, r in enumerate(rs):
        model = P.pack_shard_bits(r.read_shard())
        assert np.array_equal(g[rank, :n * 12], model[:n * 12])
        assert np.array_equal(g[rank, mask_offset:mask_offset + (n + 63) // 64 * 8], model[mask_offset:mask_offset + (n + 63) // 64 * 8])
    assert np.array_equal(P.unpack_gathered_packed(gathered.cpu().numpy(), cfg.height, cfg.width, world, band).view(np.uint32),
                          want.view(np.uint32))
    for r in (rs[0], rs[-1]):
        frame = torch.full((cfg.height, cfg.width, 4), 3.0, dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        r.unpack_gathered_packed(gathered.data_ptr(), frame.data_ptr())
        r.synchronize()
        torch.cuda.synchronize()
        assert np.array_equal(frame.cpu().numpy().view(np.uint32), want.view(np.uint32))
    for r in rs:
        r.close()
