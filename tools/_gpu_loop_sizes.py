import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from bibim_renderer_amd import Renderer, configs, textures
from oracle import bbo, scenes
maps = textures.make_material(64)
bad = 0
for it in range(int(sys.argv[1])):
    for w, h in ((1, 1), (7, 3), (65, 64), (64, 65), (130, 33), (333, 211), (16384, 33), (31, 16384)):
        sc = scenes.shaderball_scene(configs.C2.scaled(w, h, 64), bbo.MaterialData(maps))
        ref, rprim, rdepth, rst = bbo.render(sc)
        r = Renderer(w, h)
        r.render_scene(sc)
        img = r.read_framebuffer()
        prim, depth = r.read_visibility()
        img2 = r.read_framebuffer()
        st = r.stats()
        r.close()
        e1 = int((img.view(np.uint32) != ref.view(np.uint32)).any(axis=2).sum())
        e2 = int((img2.view(np.uint32) != ref.view(np.uint32)).any(axis=2).sum())
        e3 = int((prim != rprim).sum())
        if e1 or e2 or e3 or st["n_shaded"] != rst["n_shaded"]:
            bad += 1
            ys, xs = np.nonzero((img.view(np.uint32) != ref.view(np.uint32)).any(axis=2)) if e1 else np.nonzero((img2.view(np.uint32) != ref.view(np.uint32)).any(axis=2))
            print(f"iter {it} size {w}x{h}: first render {e1} px wrong, after read_visibility {e2} px wrong, prim {e3}, n_shaded {st['n_shaded']} vs {rst['n_shaded']}; rows {ys.min() if len(ys) else -1}..{ys.max() if len(ys) else -1} cols {xs.min() if len(xs) else -1}..{xs.max() if len(xs) else -1}", flush=True)
print("bad", bad)
