"""Diagnostic (needs a -DBB_ABLATE build): how much of the frame's front half (geometry + raster + items) hides behind the
shading when frames are pipelined?  Frame period of: the whole frame / only k_shade of every frame (the lists of the last whole
frame are shaded again).   usage: _gpu_overlap.py lib.so [workload] [frames_in_flight]"""
import os, sys, time, gc
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bibim_renderer_amd import _capi
_capi.LIB_PATH = os.path.abspath(sys.argv[1])
from bibim_renderer_amd import configs, textures, Renderer
from bibim_renderer_amd import scene as S
wl = sys.argv[2] if len(sys.argv) > 2 else "c3"
fif = int(sys.argv[3]) if len(sys.argv) > 3 else 3
cfg = configs.CONFIGS[wl]
r = Renderer(cfg.width, cfg.height)
r.set_option("frames_in_flight", fif)
material = r.upload_material(textures.make_material(cfg.texture_size))
scene, cam, settings = S.config_scene(r, cfg)
frames = {"c2": 600, "c3": 300, "c5": 80}[wl]
gc.collect(); gc.disable()
def rate(ablate):
    r.set_option("ablate", 0)
    for _ in range(30): S.draw_frame(r, scene, cam, settings, material)
    r.synchronize()
    r.set_option("ablate", ablate)
    for _ in range(10): S.draw_frame(r, scene, cam, settings, material)
    out = []
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(frames): S.draw_frame(r, scene, cam, settings, material)
        r.synchronize()
        out.append((time.perf_counter() - t0) / frames * 1e6)
    return out
for name, a in (("whole frame", 0), ("k_shade only", 1 << 17), ("whole frame, no lights", 2048), ("k_shade only, no lights", (1 << 17) | 2048), ("whole frame", 0)):
    print(f"{wl} fif={fif} {name:44s} us/frame " + " ".join(f"{x:7.1f}" for x in rate(a)), flush=True)
scene.close(); r.close()
