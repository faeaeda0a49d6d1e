"""Ad-hoc scale check: 256 ShaderBalls (2.5 M triangles) at 4K against the oracle, band-parallel."""
import sys, os, time
sys.path.insert(0, '.')
import numpy as np
from concurrent.futures import ThreadPoolExecutor
from dataclasses import replace
from bibim_renderer_amd import configs, textures, Renderer
from oracle import bbo, scenes
cfg = replace(configs.C5, width=3840, height=2160, grid=16, cam_pos=(0.0, 8.0, -10.0), cam_pitch=-25.0, texture_size=256, name='big')
maps = textures.make_material(256)
sc = scenes.shaderball_scene(cfg, bbo.MaterialData(maps))
r = Renderer(cfg.width, cfg.height)
t0 = time.time(); r.render_scene(sc); a = r.read_framebuffer(); st = r.stats(); print('gpu', st, f'{time.time()-t0:.2f}s')
t0 = time.time()
for _ in range(5): r.replay_frame()
r.synchronize(); print('frame', (time.time() - t0) / 5 * 1e3, 'ms')
def band(y0):
    ref, _, _, _ = bbo.render(sc, y0, min(y0 + 24, cfg.height), want_prim=False, want_depth=False)
    return y0, bool(np.array_equal(ref[y0:y0 + 24].view(np.uint32), a[y0:y0 + 24].view(np.uint32)))
t0 = time.time()
with ThreadPoolExecutor(16) as ex:
    wrong = [y0 for y0, ok in ex.map(band, range(0, cfg.height, 24)) if not ok]
print('oracle bands', f'{time.time()-t0:.1f}s', 'wrong bands:', wrong[:10])
r.close()
