import sys, ctypes as C
sys.path.insert(0, '.')
import numpy as np
from bibim_renderer_amd import configs, textures, Renderer, _capi
from bibim_renderer_amd import scene as S
_capi.LIB_PATH = 'tools/_keep/lib_stamps.so'
cfg = configs.CONFIGS[sys.argv[1]]
r = Renderer(cfg.width, cfg.height)
r.set_option('frames_in_flight', 1)
material = r.upload_material(textures.make_material(256))
scene, cam, settings = S.config_scene(r, cfg)
for _ in range(4): S.draw_frame(r, scene, cam, settings, material)
r.synchronize()
st = r.stats()
nb = (st['n_prims'] + 255) // 256
buf = np.zeros((nb, 8), np.uint64)
L = _capi.lib()
L.bbr_debug_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
rc = L.bbr_debug_stamps(r._ctx, buf.ctypes.data, nb); assert rc == 0
t = buf[:, :6].astype(np.int64)
t0 = t[:, 0].min()
rel = (t - t0) * 10 / 1e3   # wall_clock64 ticks of 10 ns -> us
print('blocks', nb, 'kernel span us', rel[:, 5].max())
names = ['start', 'draw found', 'setup done', 'bins done', 'clip done', 'end']
for b in (0, nb // 2, nb - 2, nb - 1):
    print('block', b, ' '.join(f'{n}={rel[b, i]:.2f}' for i, n in enumerate(names)))
t8 = buf.astype(np.int64); r8 = (t8 - t0) * 10 / 1e3
ok = t8[:, 6] > 0
print('setup split: loads done at +%.2f us after draw found; setup_tri done (thread 0 survivors only) at +%.2f; setup done +%.2f' % (
    (r8[ok, 6] - r8[ok, 1]).mean(), (r8[t8[:, 7] > 0, 7] - r8[t8[:, 7] > 0, 1]).mean(), (r8[:, 2] - r8[:, 1]).mean()))
d = np.diff(rel, axis=1)
print('mean phase us ', d.mean(0).round(2), 'max', d.max(0).round(2))
print('start spread', rel[:, 0].max(), 'end min/max', rel[:, 5].min(), rel[:, 5].max())
