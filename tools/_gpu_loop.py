"""Diagnostic: render N frames of a workload with a given build of the library (for rocprofv3 --pmc / --kernel-trace passes).
usage: _gpu_loop.py workload lib.so frames [frames_in_flight] [name=value ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bibim_renderer_amd import _capi
if sys.argv[2] not in ("", "-"):
    _capi.LIB_PATH = os.path.abspath(sys.argv[2])
from bibim_renderer_amd import configs, textures, Renderer
from bibim_renderer_amd import scene as S
cfg = configs.CONFIGS[sys.argv[1]]
frames = int(sys.argv[3])
r = Renderer(cfg.width, cfg.height)
r.set_option("frames_in_flight", int(sys.argv[4]) if len(sys.argv) > 4 else 1)
for o in sys.argv[5:]:
    r.set_option(o.split("=")[0], int(o.split("=")[1]))
material = r.upload_material(textures.make_material(cfg.texture_size))
scene, cam, settings = S.config_scene(r, cfg)
for _ in range(frames):
    S.draw_frame(r, scene, cam, settings, material)
r.synchronize()
scene.close(); r.close()
