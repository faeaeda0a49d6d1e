"""Diagnostic: small frames then big frames with frames in flight; report wrong pixels per read-back."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
if len(sys.argv) > 1:
    from bibim_renderer_amd import _capi
    _capi.LIB_PATH = os.path.abspath(sys.argv[1])
from bibim_renderer_amd import Renderer, configs, textures
from oracle import bbo, scenes
maps = textures.make_material(64)
big = scenes.shaderball_scene(configs.C3.scaled(960, 540, 64), bbo.MaterialData(maps))
small = scenes.triangle_scene(960, 540)
ref_big = bbo.render(big)[0]; ref_small = bbo.render(small)[0]
for layout in (0, 2):
    for fif in (1, 3):
        r = Renderer(960, 540)
        r.set_option("frames_in_flight", fif); r.set_option("stream_layout", layout)
        if os.environ.get("BIN_CAP"): r.set_option("bin_cap", int(os.environ["BIN_CAP"]))
        hb = hs = None
        for rep in range(3):
            for _ in range(4): hs = r.render_scene(small, hs)
            img = r.read_framebuffer()
            e = (img.view(np.uint32) != ref_small.view(np.uint32)).any(axis=2)
            if e.any(): print(f"layout {layout} fif {fif} rep {rep} small: {int(e.sum())} px wrong")
            for i in range(4):
                hb = r.render_scene(big, hb)
                img = r.read_framebuffer()
                e = (img.view(np.uint32) != ref_big.view(np.uint32)).any(axis=2)
                if e.any():
                    ys, xs = np.nonzero(e)
                    zero = int((img[e][:, 3] == 0).sum())
                    print(f"layout {layout} fif {fif} rep {rep} big frame {i}: {int(e.sum())} px wrong (alpha 0 in {zero}), rows {ys.min()}..{ys.max()} cols {xs.min()}..{xs.max()}")
        r.close()
print("done")
