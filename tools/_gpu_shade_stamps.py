"""Diagnostic (-DBB_STAMPS build): where a k_shade wave spends its cycles per loop iteration."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bibim_renderer_amd import _capi
_capi.LIB_PATH = os.path.abspath(sys.argv[1])
from bibim_renderer_amd import configs, textures, Renderer
from bibim_renderer_amd import scene as S
cfg = configs.CONFIGS[sys.argv[2] if len(sys.argv) > 2 else "c3"]
r = Renderer(cfg.width, cfg.height)
for o in sys.argv[3:]:
    k, v = o.split("="); r.set_option(k, int(v))
material = r.upload_material(textures.make_material(cfg.texture_size))
scene, cam, settings = S.config_scene(r, cfg)
r.set_option("frames_in_flight", 1)
for _ in range(5): S.draw_frame(r, scene, cam, settings, material)
r.synchronize()
L = C.CDLL(_capi.LIB_PATH)
buf = np.zeros((4096, 8), np.uint64)
L.bbr_debug_shade_stamps.argtypes = [C.c_void_p, C.c_void_p]
assert L.bbr_debug_shade_stamps(r._ctx, buf.ctypes.data) == 0
live = buf[buf[:, 5] > 0]
n = live[:, 5].astype(np.float64)
names = ["launch -> item, count, fragment arrived", "record (+ clip slot) arrived, varyings", "taps arrived, filter, normal", "light loop", "store"]
print(f"{len(live)} waves sampled")
for i, nm in enumerate(names):
    per = live[:, i] / n
    print(f"  {nm:48s} {per.mean():8.0f} cycles / item  (p10 {np.percentile(per,10):.0f}  p90 {np.percentile(per,90):.0f})")
print(f"  sum of phases per item {(live[:, :5].sum(1) / n).mean():.0f} cycles")
scene.close(); r.close()
