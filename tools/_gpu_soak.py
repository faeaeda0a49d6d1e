"""Soak test (GPU box): a long random walk over the asynchronous API -- scenes, extents, frames in flight, stream
layouts, presentation, fused presentation, deferred pass, caller-owned buffers -- with every read-back compared with
the oracle's frame for that (scene, extent, pass), bit for bit.
   python tools/_gpu_soak.py [seconds] [seed]"""
import os, sys, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from bibim_renderer_amd import Renderer, configs, textures
from oracle import bbo, scenes

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
mat = bbo.MaterialData(textures.make_material(64))
extents = [(320, 180), (449, 253), (640, 360)]
bases = {"c2": configs.C2, "c3": configs.C3, "c5": configs.C5}
cache = {}

def scene_and_ref(name, ext, deferred):
    key = (name, ext, deferred)
    if key not in cache:
        sc = scenes.shaderball_scene(bases[name].scaled(ext[0], ext[1], 64), mat)
        sc.frame["enable_tone_mapping"], sc.frame["exposure"] = 1, 1.25
        ref = bbo.render_deferred(sc, want_gbuffer=False)[0] if deferred else bbo.render(sc)[0]
        cache[key] = (sc, ref, bbo.present(ref, 1, 1.25))
    return cache[key]

ext = extents[0]
r = Renderer(*ext)
r.set_option("frames_in_flight", 3)
handles, deferred, fused = None, 0, 0
last = None            # (ref, ref8) of the last frame rendered into the internal frame
t_end, n_frames, n_checks, ops = time.time() + seconds, 0, 0, {}
ext_buf = None
while time.time() < t_end:
    op = rng.choice(["render"] * 12 + ["check"] * 3 + ["layout", "fif", "resize", "pass", "fused", "present", "extbuf", "sync", "route"])
    ops[op] = ops.get(op, 0) + 1
    if op == "render":
        sc, ref, ref8 = scene_and_ref(rng.choice(list(bases)), ext, deferred)
        for _ in range(rng.choice([1, 1, 2, 5, 40])):
            handles = r.render_scene(sc, handles)
            n_frames += 1
        last = (ref, ref8)
    elif op == "check" and last is not None:
        if fused:
            r.present()
            got8 = r.read_presented()
            assert np.array_equal(got8, last[1]), "fused presented image differs"
        elif ext_buf is not None:
            r.synchronize(); torch.cuda.synchronize()
            assert np.array_equal(ext_buf.cpu().numpy().view(np.uint32), last[0].view(np.uint32)), "caller buffer differs"
        else:
            assert np.array_equal(r.read_framebuffer().view(np.uint32), last[0].view(np.uint32)), "frame differs"
        n_checks += 1
    elif op == "present" and last is not None and not fused and ext_buf is None:
        r.present()
        assert np.array_equal(r.read_presented(), last[1]), "presented image differs"
        n_checks += 1
    elif op == "layout":
        r.set_option("stream_layout", rng.choice([0, 1, 2]))
    elif op == "fif":
        r.set_option("frames_in_flight", rng.choice([1, 2, 3, 4]))
    elif op == "resize":
        ext = rng.choice(extents)
        r.resize(*ext)
        last, ext_buf = None, None
    elif op == "pass":
        deferred = rng.choice([0, 1])
        r.set_option("render_pass", deferred)
        last = None
    elif op == "fused":
        fused = rng.choice([0, 1])
        r.set_option("present_fused", fused)
        last = None
        if fused:
            r.set_output_device_ptr(None, 0); ext_buf = None
    elif op == "extbuf" and not fused:
        if ext_buf is None:
            ext_buf = torch.zeros((ext[1], ext[0], 4), dtype=torch.float32, device="cuda")
            torch.cuda.synchronize()
            r.set_output_device_ptr(ext_buf.data_ptr(), ext_buf.numel() * 4)
        else:
            r.set_output_device_ptr(None, 0); ext_buf = None
        last = None
    elif op == "route":   # short / long frames' work list, k_raster's heavy tiles first or in screen order (same pixels)
        r.set_option("no_tail_items", rng.choice([0, 40000]))
        r.set_option("heavy_tiles", rng.choice([-1, 0, 4, 64]))
    elif op == "sync":
        r.synchronize()
r.synchronize()
r.close()
print(f"soak ok: {n_frames} frames, {n_checks} bit-exact checks, {len(cache)} oracle frames, ops {ops}")
