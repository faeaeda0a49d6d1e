"""Experiment (GPU box): host cost of submitting one frame -- a 64x64 frame of the C2 scene keeps the GPU idle, so
the wall time per frame is what the host spends in bb::drawFrame (Python -> ctypes -> C++ shim -> HIP calls).
   python tools/_gpu_host_cost.py"""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bibim_renderer_amd import Renderer, configs, textures
from bibim_renderer_amd import scene as S

maps = textures.make_material(64)
ball = S.load_shaderball_vertices()
cfg = configs.C2.scaled(64, 64, 64)
for layout in (0, 1, 2):
    for timing in (0, 2, 1):
        r = Renderer(cfg.width, cfg.height)
        r.set_option("frames_in_flight", 3)
        r.set_option("stream_layout", layout)
        mat = r.upload_material(maps)
        scene, cam, settings = S.config_scene(r, cfg, ball)
        for _ in range(50):
            S.draw_frame(r, scene, cam, settings, mat)
        r.synchronize()
        r.set_option("timing", timing)
        gc.collect(); gc.disable()
        n = 2000
        t0 = time.perf_counter()
        for _ in range(n):
            S.draw_frame(r, scene, cam, settings, mat)
        t1 = time.perf_counter()
        r.synchronize()
        t2 = time.perf_counter()
        gc.enable()
        print(f"layout {layout} timing {timing}: {1e6 * (t1 - t0) / n:6.1f} us per frame submitted, {1e6 * (t2 - t0) / n:6.1f} us incl. drain", flush=True)
        scene.close(); r.close()
