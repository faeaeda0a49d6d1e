"""Diagnostic: why bench.py's 20-step driver-style run reads higher than the same 20 frames in tools/_gpu_rate.py.
Variants of the closing synchronisation and of what is loaded in the process.   usage: _gpu_drv_style.py"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os, time, gc
sys.path.insert(0, %(root)r)
%(pre)s
from bibim_renderer_amd import configs, textures, Renderer
from bibim_renderer_amd import scene as S
cfg = configs.CONFIGS["c3"]
r = Renderer(cfg.width, cfg.height)
r.set_option("frames_in_flight", 3)
material = r.upload_material(textures.make_material(cfg.texture_size))
scene, cam, settings = S.config_scene(r, cfg)
S.draw_frame(r, scene, cam, settings, material); r.synchronize()
gc.collect(); gc.disable()
for _ in range(40): S.draw_frame(r, scene, cam, settings, material)
r.synchronize()
%(opt)s
out = []
for rep in range(6):
    for _ in range(5): S.draw_frame(r, scene, cam, settings, material)
    %(sync)s
    t0 = time.perf_counter()
    for _ in range(20): S.draw_frame(r, scene, cam, settings, material)
    %(sync)s
    out.append((time.perf_counter() - t0) / 20 * 1e6)
print("%(tag)-58s 20-step " + " ".join("%%7.1f" %% x for x in out), flush=True)
scene.close(); r.close()
'''
cases = [("plain: r.synchronize()", "", "", "r.synchronize()"),
         ("import torch; r.synchronize()", "import torch", "", "r.synchronize()"),
         ("import torch; torch.cuda.synchronize()", "import torch; torch.cuda.init()", "", "torch.cuda.synchronize()"),
         ("import torch; torch sync; timing=2 stride 4", "import torch; torch.cuda.init()", "r.set_option('timing', 2); r.set_option('timing_stride', 4)", "torch.cuda.synchronize()"),
         ("plain again", "", "", "r.synchronize()")]
for tag, pre, opt, sync in cases:
    code = CHILD % dict(root=ROOT, pre=pre, opt=opt, sync=sync, tag=tag)
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    sys.stdout.write(p.stdout)
    if p.returncode:
        sys.stdout.write(f"{tag}: FAILED\n{p.stderr[-800:]}\n")
    sys.stdout.flush()
