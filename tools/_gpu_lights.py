import sys
sys.path.insert(0, '.')
from bibim_renderer_amd import configs, textures, Renderer
from bibim_renderer_amd import scene as S
cfg = configs.C3
maps = textures.make_material(2048)
r = Renderer(cfg.width, cfg.height)
r.set_option('frames_in_flight', 1)
material = r.upload_material(maps)
scene, cam, settings = S.config_scene(r, cfg)
S.draw_frame(r, scene, cam, settings, material); r.synchronize()
r.set_option('timing', 1)
def run(tag):
    for _ in range(5): S.draw_frame(r, scene, cam, settings, material)
    r.timing_reset()
    for _ in range(30): S.draw_frame(r, scene, cam, settings, material)
    n, f, g, ra, t = r.timing_summary()
    print(f'{tag:28s} frame {f*1e3:8.1f} us  geometry {g*1e3:7.1f}  raster {ra*1e3:7.1f}  shade {t*1e3:7.1f}')
for nl in (0, 1, 2, 4, 8, 16):
    scene.set_point_lights(configs._grid_lights(nl, 2) if nl else [])
    run(f'{nl} lights')
scene.set_point_lights(cfg.lights)
for ab, name in ((8, 'uniform uv'), (16, 'uniform attrs'), (24, 'uniform uv+attrs'), (4, 'no plane'), (4+24, 'no plane, uniform uv+attrs')):
    r.set_option('ablate', ab); run(name)
r.set_option('ablate', 0)
settings.enable_normal_map = 0; run('normal map off')
