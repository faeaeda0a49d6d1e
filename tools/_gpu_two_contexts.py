"""Diagnostic: would MORE frames in flight than a context's four slots help a short (1080p) frame?  Two contexts of one process
render the same workload alternately (eight frames in flight, eight streams); aggregate us per frame against one context.
usage: [GPU_MAX_HW_QUEUES=8] python tools/_gpu_two_contexts.py [c2]"""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bibim_renderer_amd import configs, textures, Renderer
from bibim_renderer_amd import scene as S
wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
cfg = configs.CONFIGS[wl]
maps = textures.make_material(cfg.texture_size)
ctx = []
for _ in range(2):
    r = Renderer(cfg.width, cfg.height)
    r.set_option("frames_in_flight", 4)
    m = r.upload_material(maps)
    sc, cam, st = S.config_scene(r, cfg)
    S.draw_frame(r, sc, cam, st, m); r.synchronize()
    ctx.append((r, sc, cam, st, m))
gc.collect(); gc.disable()
def run(n_ctx, frames=800):
    use = ctx[:n_ctx]
    for _ in range(100):
        for (r, sc, cam, st, m) in use: S.draw_frame(r, sc, cam, st, m)
    for (r, *_rest) in use: r.synchronize()
    out = []
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(frames // n_ctx):
            for (r, sc, cam, st, m) in use: S.draw_frame(r, sc, cam, st, m)
        for (r, *_rest) in use: r.synchronize()
        out.append((time.perf_counter() - t0) / (frames // n_ctx * n_ctx) * 1e6)
    return out
print(f"{wl} GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES', '(default)')}: one context, 4 in flight: us/frame " + " ".join(f"{x:6.1f}" for x in run(1)), flush=True)
print(f"{wl} two contexts alternately, 8 in flight:          us/frame " + " ".join(f"{x:6.1f}" for x in run(2)), flush=True)
print(f"{wl} one context again:                              us/frame " + " ".join(f"{x:6.1f}" for x in run(1)), flush=True)
for (r, sc, *_r) in ctx: sc.close(); r.close()
