"""Render N frames of a BASELINE config through the host shim + C ABI (no torch, no oracle): the command
rocprofv3 wraps for kernel traces and PMC passes."""
import argparse, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if "--lib" in sys.argv:  # diagnostic: another build of the library
    from bibim_renderer_amd import _capi
    _capi.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
from bibim_renderer_amd import configs, textures, Renderer
from bibim_renderer_amd import scene as S

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="c3"); ap.add_argument("--frames", type=int, default=20)
ap.add_argument("--tile-mode", type=int, default=0); ap.add_argument("--opt", action="append", default=[]); ap.add_argument("--lib", default=None)
a = ap.parse_args()
cfg = configs.CONFIGS[a.workload]
r = Renderer(cfg.width, cfg.height)
r.set_option("tile_mode", a.tile_mode)
for o in a.opt:
    k, v = o.split("="); r.set_option(k, int(v))
material = r.upload_material(textures.make_material(cfg.texture_size))
scene, cam, settings = S.config_scene(r, cfg)
for _ in range(a.frames):
    S.draw_frame(r, scene, cam, settings, material)
r.synchronize()
print(r.stats())
scene.close(); r.close()
