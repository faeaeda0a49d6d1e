import sys, time
sys.path.insert(0, '.')
import numpy as np
from oracle import bbo, scenes
from bibim_renderer_amd import configs, textures, Renderer

def cmp(name, cfg, texsize):
    mat = bbo.MaterialData(textures.make_material(texsize))
    sc = scenes.shaderball_scene(cfg, mat)
    t = time.time(); ref, rprim, rdepth, st = bbo.render(sc); t_o = time.time() - t
    r = Renderer(cfg.width, cfg.height)
    h = r.render_scene(sc)
    img = r.read_framebuffer()
    gs = r.stats()
    prim, depth = r.read_visibility()
    print(name, 'oracle', st, f'{t_o:.2f}s')
    print(name, 'gpu   ', gs)
    print(' prim mismatches', int((prim != rprim).sum()), 'depth mismatches', int((depth != rdepth).sum()))
    d = np.abs(img - ref); tol = 1e-4 * np.maximum(1, np.abs(ref))
    print(' max abs diff', float(d.max()), 'violations', int((d > tol).sum()), 'bit-exact pixels', float((img.view(np.uint32) == ref.view(np.uint32)).all(-1).mean()))
    nan = np.isnan(img).sum(), np.isnan(ref).sum()
    print(' nans', nan)
    r.set_option('timing', 1)
    for _ in range(3): r.replay_frame()
    r.synchronize()
    print(' frame ms, tile ms', r.last_frame_time_ms())
    r.close()

cmp('C2@480x270', configs.C2.scaled(480, 270, 256), 256)
cmp('C3@960x540', configs.C3.scaled(960, 540, 512), 512)
cmp('C2 full', configs.C2, 2048)
