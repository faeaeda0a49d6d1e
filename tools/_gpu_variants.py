"""Diagnostic: time the kernels of one workload, each alone on the GPU (one frame in flight, HIP events around every
kernel), for several builds of the library.  usage: _gpu_variants.py [--workload c3] lib1.so [lib2.so ...]
Each build runs in its own process (the library path is patched before the first load)."""
import argparse, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os
sys.path.insert(0, %(root)r)
from bibim_renderer_amd import _capi
_capi.LIB_PATH = %(lib)r
from bibim_renderer_amd import configs, textures, Renderer
from bibim_renderer_amd import scene as S
cfg = configs.CONFIGS[%(workload)r]
r = Renderer(cfg.width, cfg.height)
for k, v in %(opts)r: r.set_option(k, v)
material = r.upload_material(textures.make_material(cfg.texture_size))
scene, cam, settings = S.config_scene(r, cfg)
for _ in range(10): S.draw_frame(r, scene, cam, settings, material)
r.synchronize()
r.set_option("frames_in_flight", 1); r.set_option("timing_stride", 1); r.set_option("timing", 1)
for _ in range(5): S.draw_frame(r, scene, cam, settings, material)
r.synchronize(); r.timing_reset()
for _ in range(%(frames)d): S.draw_frame(r, scene, cam, settings, material)
r.synchronize()
n, f, g, ra, s = r.timing_summary()
print("%(tag)s: frames %%d  frame latency %%.1f us  geometry %%.1f  raster %%.1f  shade %%.1f us" %% (n, f * 1e3, g * 1e3, ra * 1e3, s * 1e3))
scene.close(); r.close()
'''
ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="c3"); ap.add_argument("--frames", type=int, default=40)
ap.add_argument("--opt", action="append", default=[]); ap.add_argument("libs", nargs="+")
a = ap.parse_args()
opts = [(o.split("=")[0], int(o.split("=")[1])) for o in a.opt]
rc = 0
for lib in a.libs:
    code = CHILD % dict(root=ROOT, lib=os.path.abspath(lib), workload=a.workload, opts=opts, frames=a.frames, tag=os.path.basename(lib))
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    sys.stdout.write(p.stdout)
    if p.returncode:
        sys.stdout.write(f"{lib}: FAILED rc={p.returncode}\n{p.stderr[-2000:]}\n"); rc = 1
    sys.stdout.flush()
sys.exit(rc)
