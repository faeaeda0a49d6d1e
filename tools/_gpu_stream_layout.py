"""Experiment (GPU box): frame rate with the three stream layouts of option stream_layout, over resolutions and
light counts -- where does the geometry -> raster chain, rather than the GPU, set the rate?
   python tools/_gpu_stream_layout.py"""
import gc, os, sys, time
from dataclasses import replace
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bibim_renderer_amd import Renderer, configs, textures
from bibim_renderer_amd import scene as S

maps = textures.make_material(2048)
ball = S.load_shaderball_vertices()
cases = []
for base, (w, h) in [(configs.C2, (1920, 1080)), (configs.C2, (3840, 2160)), (configs.C3, (1920, 1080)), (configs.C3, (2560, 1440)),
                     (configs.C3, (3840, 2160)), (configs.C5, (3840, 2160)), (configs.C5, (7680, 4320))]:
    for nl in sorted({1, len(base.lights), 8}):
        lights = (base.lights * 8)[:nl]
        cases.append(replace(base, width=w, height=h, lights=tuple(lights), name=f"{base.name}@{w}x{h}/L{nl}"))
if os.environ.get("QUICK"):
    cases = [c for c in cases if c.name in os.environ["QUICK"].split(",")]
for cfg in cases:
    res = []
    for mode in (0, 1, 2):
        r = Renderer(cfg.width, cfg.height)
        r.set_option("frames_in_flight", int(os.environ.get("FIF", "3")))
        r.set_option("stream_layout", mode)
        mat = r.upload_material(maps)
        scene, cam, settings = S.config_scene(r, cfg, ball)
        S.draw_frame(r, scene, cam, settings, mat); r.synchronize()
        n = 100 if cfg.width > 4000 else 300
        for _ in range(20):
            S.draw_frame(r, scene, cam, settings, mat)
        r.synchronize()
        gc.collect(); gc.disable()   # a gen-2 collection of torch's object graph is a 40 ms host stall
        t0 = time.perf_counter()
        worst, worst_i, tp = 0.0, -1, t0
        for i in range(n):
            S.draw_frame(r, scene, cam, settings, mat)
            tn = time.perf_counter()
            if tn - tp > worst:
                worst, worst_i = tn - tp, i
            tp = tn
        r.synchronize()
        res.append((time.perf_counter() - t0) / n * 1e6)
        gc.enable()
        if os.environ.get("QUICK"):
            print(f"   mode {mode}: slowest submit {worst * 1e3:.2f} ms at frame {worst_i}, capacity retries {r.stats()['bin_overflow']}", flush=True)
        scene.close(); r.close()
    load = cfg.width * cfg.height * max(len(cfg.lights), 1) / 1e6
    print(f"{cfg.name:28s} load {load:7.1f}M  shared {res[0]:8.1f} us  own {res[1]:8.1f} us ({res[1] / res[0]:.3f})  per-slot {res[2]:8.1f} us ({res[2] / res[0]:.3f})", flush=True)
