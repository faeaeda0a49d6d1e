import sys, time
sys.path.insert(0, '.')
import numpy as np
from bibim_renderer_amd import configs, textures, Renderer
from bibim_renderer_amd import scene as S
cfg = configs.C3
maps = textures.make_material(2048)
for tile_mode in (1,):
    r = Renderer(cfg.width, cfg.height)
    r.set_option('tile_mode', tile_mode); r.set_option('frames_in_flight', 1)
    material = r.upload_material(maps)
    scene, cam, settings = S.config_scene(r, cfg)
    S.draw_frame(r, scene, cam, settings, material); r.synchronize()
    print('tile_mode', tile_mode, r.stats())
    r.set_option('timing', 1)
    for ab, name in ((0, 'full'), (2, 'no shading'), (2+32, 'geom: no binning'), (2+64, 'geom: no stores'), (2+128, 'geom: no clip'), (2+32+64+128, 'geom: none of them'), (2+256, 'raster: no class0'), (2+512, 'raster: no class1'), (2+1024, 'raster: no class2'), (2+4, 'raster: no broad'), (2+256+512+1024+4, 'raster: nothing')):
        r.set_option('ablate', ab)
        for _ in range(5): S.draw_frame(r, scene, cam, settings, material)
        r.timing_reset()
        for _ in range(20): S.draw_frame(r, scene, cam, settings, material)
        n, f, g, ra, t = r.timing_summary()
        print(f'  {name:24s} frame {f*1e3:8.1f} us  geometry {g*1e3:8.1f} us  raster {ra*1e3:8.1f} us  shade {t*1e3:8.1f} us')
    r.set_option('ablate', 0)
    scene.close(); r.close()
