import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bibim_renderer_amd import Renderer, configs, textures
from oracle import bbo, scenes
maps = textures.make_material(64)
big = scenes.shaderball_scene(configs.C3.scaled(960, 540, 64), bbo.MaterialData(maps))
ref_big = bbo.render(big)[0]
r = Renderer(960, 540)
r.set_option("frames_in_flight", 1)
h = None
for i in range(4):
    h = r.render_scene(big, h)
    img = r.read_framebuffer()
    e = (img.view(np.uint32) != ref_big.view(np.uint32)).any(axis=2)
    print(f"frame {i}: {int(e.sum())} px wrong; stats {r.stats()}", flush=True)
    img = r.read_framebuffer()
    e = (img.view(np.uint32) != ref_big.view(np.uint32)).any(axis=2)
    print(f"   read again: {int(e.sum())} px wrong", flush=True)
r.close()
