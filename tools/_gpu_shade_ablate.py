import sys
sys.path.insert(0, '.')
from bibim_renderer_amd import configs, textures, Renderer
from bibim_renderer_amd import scene as S
cfg = configs.C3
for tex in (2048, 256):
    r = Renderer(cfg.width, cfg.height)
    r.set_option('frames_in_flight', 1)
    material = r.upload_material(textures.make_material(tex))
    scene, cam, settings = S.config_scene(r, cfg)
    S.draw_frame(r, scene, cam, settings, material); r.synchronize()
    r.set_option('timing', 1)
    for ab, name in ((0, 'full'), (8, 'uniform uv'), (16, 'prim 0 attrs'), (24, 'both')):
        r.set_option('ablate', ab)
        for _ in range(5): S.draw_frame(r, scene, cam, settings, material)
        r.timing_reset()
        for _ in range(20): S.draw_frame(r, scene, cam, settings, material)
        n, f, g, ra, t = r.timing_summary()
        print(f'tex {tex:5d} {name:16s} frame {f*1e3:8.1f} us  geometry {g*1e3:8.1f} raster {ra*1e3:8.1f} shade {t*1e3:8.1f}')
    r.set_option('ablate', 0)
    scene.close(); r.close()
