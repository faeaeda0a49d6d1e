"""Diagnostic (GPU box): the PIPELINED frame period of one rank of a screen-band partition -- three frames in flight, no
exchange: what a rank can render per second before a byte moves -- for world 1 / 2 / 4 / 8, every rank index, and two band
heights.  Beside it each kernel of that rank alone (one frame in flight, HIP events).  The 1 / world ideal is printed for
comparison.   usage: _gpu_band_throughput.py [--workload c3] [--band-rows 0 256]   -> profiles/r05_band_throughput.txt"""
import argparse, gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bibim_renderer_amd import configs, textures, Renderer
from bibim_renderer_amd import scene as S

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="c3"); ap.add_argument("--frames", type=int, default=200)
ap.add_argument("--band-rows", type=int, nargs="+", default=[0, 256])
a = ap.parse_args()
cfg = configs.CONFIGS[a.workload]
maps = textures.make_material(cfg.texture_size)
base = None
print(f"# {a.workload}: us per frame of ONE rank's share, three frames in flight, {a.frames} frames, no exchange; then its kernels alone", flush=True)
for band in a.band_rows:
    for world in (1, 2, 4, 8):
        if world == 1 and band != a.band_rows[0]:
            continue
        worst = 0.0
        for rank in range(world):
            r = Renderer(cfg.width, cfg.height)
            r.set_option("frames_in_flight", 3)
            if world > 1:
                r.set_partition(rank, world, band)
            material = r.upload_material(maps)
            scene, cam, settings = S.config_scene(r, cfg)
            S.draw_frame(r, scene, cam, settings, material); r.synchronize()
            gc.collect(); gc.disable()
            for _ in range(40):
                S.draw_frame(r, scene, cam, settings, material)
            r.synchronize()
            best = 1e9
            for rep in range(2):
                t0 = time.perf_counter()
                for _ in range(a.frames):
                    S.draw_frame(r, scene, cam, settings, material)
                r.synchronize()
                best = min(best, (time.perf_counter() - t0) / a.frames * 1e6)
            r.set_option("frames_in_flight", 1); r.set_option("timing_stride", 1); r.set_option("timing", 1)
            for _ in range(5):
                S.draw_frame(r, scene, cam, settings, material)
            r.synchronize(); r.timing_reset()
            for _ in range(30):
                S.draw_frame(r, scene, cam, settings, material)
            r.synchronize()
            n, f, g, ra, s = r.timing_summary()
            st = r.stats()
            shard_rows = r.shard_rows() if world > 1 else cfg.height
            gc.enable()
            if world == 1:
                base = best
            worst = max(worst, best)
            print(f"band rows {band or r.tile_height():3d} world {world} rank {rank}: pipelined {best:7.1f} us/frame   alone: geometry {g * 1e3:5.1f} "
                  f"raster {ra * 1e3:5.1f} shade {s * 1e3:5.1f} chain {f * 1e3:6.1f} us   shaded {st['n_shaded']}", flush=True)
            scene.close(); r.close()
        if world > 1 and base:
            print(f"  -> world {world}, band rows {band or 32}: slowest rank {worst:6.1f} us/frame = {base / worst:4.2f}x one GPU's {base:5.1f} "
                  f"(ideal {world}x = {base / world:5.1f} us)", flush=True)
            # what the links allow for the frame this rendering would have to be put together into: every rank receives world - 1
            # blocks, each over its own xGMI link at best (76.8 GB/s per direction) -- block / link rate; a ring: x (world - 1)
            px = cfg.width * shard_rows
            forms = (("rgba32f", 16.0), ("packed rgb + alpha bit", 12.0 + 1.0 / 8), ("rgba16f (lossy)", 8.0), ("rgba8 presented", 4.0))
            print("     exchange, link-bound (direct / ring) per frame: " + "; ".join(
                f"{name} {px * b / 76.8e9 * 1e6:6.0f} / {px * b / 76.8e9 * 1e6 * (world - 1):6.0f} us" for name, b in forms) +
                f"   -> which forms can beat one GPU's {base:5.1f} us at all: " +
                (", ".join(name for name, b in forms if px * b / 76.8e9 * 1e6 < base) or "none"), flush=True)
