"""Diagnostic (GPU box): what ONE rank of a screen-band partition costs per kernel, each kernel alone on the GPU (one frame in
flight, HIP events): rank 0 of world 1 / 2 / 4 / 8 of a workload.  With band-aware geometry a rank writes records and bin
entries only for primitives that touch its bands; its vertex positions are still everybody's.
usage: _gpu_band_kernels.py [--workload c3] [--frames 40]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bibim_renderer_amd import configs, textures, Renderer
from bibim_renderer_amd import scene as S

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="c3"); ap.add_argument("--frames", type=int, default=40)
a = ap.parse_args()
cfg = configs.CONFIGS[a.workload]
maps = textures.make_material(cfg.texture_size)
for world in (1, 2, 4, 8):
    for rank in sorted({0, world - 1}):
        r = Renderer(cfg.width, cfg.height)
        if world > 1:
            r.set_partition(rank, world, 0)
        material = r.upload_material(maps)
        scene, cam, settings = S.config_scene(r, cfg)
        for _ in range(6):
            S.draw_frame(r, scene, cam, settings, material)
        r.synchronize()
        r.set_option("frames_in_flight", 1); r.set_option("timing_stride", 1); r.set_option("timing", 1)
        for _ in range(5):
            S.draw_frame(r, scene, cam, settings, material)
        r.synchronize(); r.timing_reset()
        for _ in range(a.frames):
            S.draw_frame(r, scene, cam, settings, material)
        r.synchronize()
        n, f, g, ra, s = r.timing_summary()
        st = r.stats()
        print(f"{a.workload} world {world} rank {rank}: frame latency {f * 1e3:7.1f} us  geometry {g * 1e3:6.1f}  raster {ra * 1e3:6.1f}  "
              f"shade {s * 1e3:6.1f} us   raster tris {st.get('n_raster_tris')}  shaded {st['n_shaded']}", flush=True)
        scene.close(); r.close()
