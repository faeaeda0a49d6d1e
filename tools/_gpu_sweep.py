import sys, itertools
sys.path.insert(0, '.')
from bibim_renderer_amd import configs, textures, Renderer
from bibim_renderer_amd import scene as S
cfg = configs.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else 'c3']
opt = sys.argv[2]; values = [int(v) for v in sys.argv[3].split(',')]
maps = textures.make_material(cfg.texture_size)
r = Renderer(cfg.width, cfg.height)
r.set_option('frames_in_flight', 1)
material = r.upload_material(maps)
scene, cam, settings = S.config_scene(r, cfg)
S.draw_frame(r, scene, cam, settings, material); r.synchronize()
r.set_option('timing', 1)
for v in values:
    r.set_option(opt, v)
    for _ in range(5): S.draw_frame(r, scene, cam, settings, material)
    r.timing_reset()
    for _ in range(30): S.draw_frame(r, scene, cam, settings, material)
    n, f, g, ra, t = r.timing_summary()
    print(f'{opt}={v:8d}  frame {f*1e3:8.1f} us  geometry {g*1e3:8.1f}  raster {ra*1e3:8.1f}  shade {t*1e3:8.1f}')
