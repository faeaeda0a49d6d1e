import sys, time
sys.path.insert(0, '.')
from bibim_renderer_amd import configs, textures, Renderer
from bibim_renderer_amd import scene as S
maps = textures.make_material(2048)
for name in ('c2', 'c3', 'c5'):
    cfg = configs.CONFIGS[name]
    for fif in (1, 2, 3):
        r = Renderer(cfg.width, cfg.height)
        r.set_option('frames_in_flight', fif)
        for o in sys.argv[1:]:
            k, v = o.split('='); r.set_option(k, int(v))
        material = r.upload_material(maps)
        scene, cam, settings = S.config_scene(r, cfg)
        S.draw_frame(r, scene, cam, settings, material); r.synchronize()
        st = r.stats()
        r.set_option('timing', 1)
        for _ in range(5): S.draw_frame(r, scene, cam, settings, material)
        r.synchronize(); r.timing_reset()
        t0 = time.perf_counter()
        K = 50
        for _ in range(K): S.draw_frame(r, scene, cam, settings, material)
        r.synchronize()
        dt = (time.perf_counter() - t0) / K
        n, f, g, ra, t = r.timing_summary()
        print(f'{name} fif={fif} step {dt*1e6:8.1f} us = {cfg.width*cfg.height/dt/1e9:6.2f} Gpix/s | geometry {g*1e3:7.1f} raster {ra*1e3:7.1f} shade {t*1e3:7.1f} | prims {st["n_prims"]} raster_tris {st["n_raster_tris"]} refs {st["n_bin_refs"]} broad {st["n_broad_tris"]} shaded {st["n_shaded"]} retries {st["bin_overflow"]}')
        scene.close(); r.close()
