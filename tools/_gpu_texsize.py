import sys, os, time, gc
sys.path.insert(0, os.getcwd())
from bibim_renderer_amd import configs, textures, Renderer
from bibim_renderer_amd import scene as S
for name, tex in (("c3", 2048), ("c3", 256), ("c3", 64), ("c5", 2048), ("c5", 64)):
    base = configs.CONFIGS[name]
    cfg = base.scaled(base.width, base.height, tex)
    r = Renderer(cfg.width, cfg.height); r.set_option("frames_in_flight", 3)
    material = r.upload_material(textures.make_material(tex))
    scene, cam, settings = S.config_scene(r, cfg)
    S.draw_frame(r, scene, cam, settings, material); r.synchronize()
    for _ in range(300): S.draw_frame(r, scene, cam, settings, material)
    r.synchronize()
    out = []
    n = 300 if name == "c3" else 80
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(n): S.draw_frame(r, scene, cam, settings, material)
        r.synchronize(); out.append((time.perf_counter() - t0) / n * 1e6)
    r.set_option("frames_in_flight", 1); r.set_option("timing_stride", 1); r.set_option("timing", 1)
    for _ in range(5): S.draw_frame(r, scene, cam, settings, material)
    r.synchronize(); r.timing_reset()
    for _ in range(30): S.draw_frame(r, scene, cam, settings, material)
    r.synchronize()
    _, f, g, ra, s = r.timing_summary()
    print(f"{name} maps {tex}^2: us/frame " + " ".join(f"{x:.1f}" for x in out) + f"   alone: geometry {g*1e3:.1f} raster {ra*1e3:.1f} shade {s*1e3:.1f}", flush=True)
    scene.close(); r.close()
