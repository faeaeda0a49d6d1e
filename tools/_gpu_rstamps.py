import sys, ctypes as C
sys.path.insert(0, '.')
import numpy as np
from bibim_renderer_amd import configs, textures, Renderer, _capi
from bibim_renderer_amd import scene as S
import os; _capi.LIB_PATH = os.path.abspath(os.environ.get('BBR_STAMPS_LIB', 'tools/_keep/stamps.so'))
cfg = configs.CONFIGS[sys.argv[1]]
r = Renderer(cfg.width, cfg.height)
r.set_option('frames_in_flight', 1)
if len(sys.argv) > 2: r.set_option('ablate', int(sys.argv[2]))
material = r.upload_material(textures.make_material(cfg.texture_size))
scene, cam, settings = S.config_scene(r, cfg)
for _ in range(4): S.draw_frame(r, scene, cam, settings, material)
r.synchronize()
st = r.stats()
nt = st['n_tiles']
buf = np.zeros((nt, 8), np.uint64)
L = _capi.lib()
L.bbr_debug_raster_stamps.argtypes = [C.c_void_p, C.c_void_p]
assert L.bbr_debug_raster_stamps(r._ctx, buf.ctypes.data) == 0
t = buf[:, :6].astype(np.int64); info = buf[:, 6]
n_entries = (info >> np.uint64(32)).astype(np.int64); n_cov = (info & np.uint64(0xFFFFFFFF)).astype(np.int64)
cls = np.stack([(buf[:, 7] >> np.uint64(16 * k)) & np.uint64(0xFFFF) for k in range(4)], 1).astype(np.int64)
t0 = t[:, 0].min()
rel = (t - t0) * 10 / 1e3
life = rel[:, 4] - rel[:, 0]
print('tiles', nt, 'kernel span us', rel[:, 4].max(), 'mean WG life', life.mean(), 'max', life.max())
d = np.diff(rel[:, :5], axis=1)
names = ['init+counts', 'stage', 'raster', 'compact']
for lo, hi, tag in ((0, 0, 'empty'), (1, 8, '1-8 entries'), (9, 64, '9-64'), (65, 256, '65-256'), (257, 10**9, '>256')):
    m = (n_entries >= lo) & (n_entries <= hi)
    if m.sum() == 0: continue
    print(f'   entries by class (tiny, small, large, every-tile): mean {cls[m].mean(0).round(1).tolist()} max {cls[m].max(0).tolist()}')
    print(f'{tag:12s} tiles {int(m.sum()):5d} life {life[m].mean():6.2f} us  phases ' + ' '.join(f'{n}={d[m, i].mean():.2f}' for i, n in enumerate(names)) + f'  cov {n_cov[m].mean():.0f}')
# concurrency: average number of WGs alive
ev = np.concatenate([np.stack([rel[:, 0], np.ones(nt)], 1), np.stack([rel[:, 4], -np.ones(nt)], 1)])
ev = ev[np.argsort(ev[:, 0])]
alive = np.cumsum(ev[:, 1]); dt = np.diff(ev[:, 0]); print('avg WGs alive', (alive[:-1] * dt).sum() / dt.sum(), 'start of last WG', rel[:, 0].max())
# timeline: WGs alive, started and finished per 2-us slice
edges = np.arange(0, rel[:, 4].max() + 2, 2.0)
started = np.histogram(rel[:, 0], edges)[0]; finished = np.histogram(rel[:, 4], edges)[0]
alive_at = [(int(((rel[:, 0] <= t) & (rel[:, 4] > t)).sum())) for t in edges[:-1]]
print('t_us   alive started finished')
for t, a, s_, f_ in zip(edges[:-1], alive_at, started, finished): print(f'{t:5.0f} {a:6d} {s_:7d} {f_:8d}')
