"""Diagnostic: pipelined frame period of a workload for several builds / option sets, without bench.py's extras (no torch,
no oracle): warm-up, then N frames between two synchronisations.  Each case runs in its own process.
usage: _gpu_rate.py [--reps 3] CASE [CASE ...]     CASE = workload[:lib.so][:fif=3][:name=value ...]   e.g. c3::fif=2:stream_layout=1"""
import argparse, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os, time, gc
sys.path.insert(0, %(root)r)
from bibim_renderer_amd import _capi
if %(lib)r: _capi.LIB_PATH = %(lib)r
from bibim_renderer_amd import configs, textures, Renderer
from bibim_renderer_amd import scene as S
cfg = configs.CONFIGS[%(workload)r]
r = Renderer(cfg.width, cfg.height)
r.set_option("frames_in_flight", %(fif)d)
for k, v in %(opts)r: r.set_option(k, v)
material = r.upload_material(textures.make_material(cfg.texture_size))
scene, cam, settings = S.config_scene(r, cfg)
S.draw_frame(r, scene, cam, settings, material); r.synchronize()
gc.collect(); gc.disable()
for _ in range(60): S.draw_frame(r, scene, cam, settings, material)
r.synchronize()
out = []
for rep in range(%(reps)d):
    for _ in range(10): S.draw_frame(r, scene, cam, settings, material)
    t0 = time.perf_counter()
    for _ in range(%(frames)d): S.draw_frame(r, scene, cam, settings, material)
    r.synchronize()
    out.append((time.perf_counter() - t0) / %(frames)d * 1e6)
# short run, driver style: 20 frames from a drained GPU
drv = []
for rep in range(3):
    for _ in range(5): S.draw_frame(r, scene, cam, settings, material)
    r.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): S.draw_frame(r, scene, cam, settings, material)
    r.synchronize()
    drv.append((time.perf_counter() - t0) / 20 * 1e6)
print("%(tag)-44s us/frame " + " ".join("%%7.1f" %% x for x in out) + "   20-step " + " ".join("%%7.1f" %% x for x in drv), flush=True)
scene.close(); r.close()
'''
ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("cases", nargs="+")
a = ap.parse_args()
rc = 0
for case in a.cases:
    parts = case.split(":")
    workload = parts[0]
    lib = parts[1] if len(parts) > 1 else ""
    fif, opts = (4 if workload == "c2" else 3), []
    for p in parts[2:]:
        k, v = p.split("=")
        if k == "fif":
            fif = int(v)
        else:
            opts.append((k, int(v)))
    frames = {"c2": 600, "c3": 300, "c5": 80}[workload]
    code = CHILD % dict(root=ROOT, lib=os.path.abspath(lib) if lib else "", workload=workload, opts=opts, frames=frames, fif=fif,
                        reps=a.reps, tag=case)
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    sys.stdout.write(p.stdout)
    if p.returncode:
        sys.stdout.write(f"{case}: FAILED rc={p.returncode}\n{p.stderr[-1500:]}\n"); rc = 1
    sys.stdout.flush()
sys.exit(rc)
