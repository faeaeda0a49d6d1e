"""Stress rehearsal: one rank of a screen-band partition rendering into two alternating caller buffers (as bench.py's
N > 1 path does), every frame's shard copied aside on a second stream; all copies must be identical."""
import sys, os
sys.path.insert(0, '.')
import numpy as np, torch
from bibim_renderer_amd import configs, textures, Renderer
from bibim_renderer_amd import scene as S
fif = int(sys.argv[1]); steps = int(sys.argv[2]); present = len(sys.argv) > 3 and sys.argv[3] == 'present'
world, rank = 4, int(os.environ.get('STRESS_RANK', '1'))
cfg = configs.CONFIGS[os.environ.get('STRESS_CFG', 'c3')]
r = Renderer(cfg.width, cfg.height)
r.set_option('frames_in_flight', fif)
for kv in os.environ.get('STRESS_OPTS', '').split():
    k, v = kv.split('='); r.set_option(k, int(v))
material = r.upload_material(textures.make_material(cfg.texture_size))
scene, cam, settings = S.config_scene(r, cfg)
r.set_partition(rank, world, r.tile_height())
rows, W = r.shard_rows(), cfg.width
shard = [torch.empty((rows, W, 4), dtype=torch.float32, device='cuda') for _ in range(2)]
shard8 = [torch.empty((rows, W, 4), dtype=torch.uint8, device='cuda') for _ in range(2)]
keep = torch.empty((steps, rows, W, 4), dtype=torch.uint8 if present else torch.float32, device='cuda')
ag = torch.cuda.Stream(); consumed = [torch.cuda.Event(), torch.cuda.Event()]
for n in range(steps):
    b = n & 1
    if n >= 2: r.wait_event(consumed[b].cuda_event)
    r.set_output_device_ptr(shard[b].data_ptr(), shard[b].numel() * 4)
    S.draw_frame(r, scene, cam, settings, material)
    if present: r.present(shard8[b].data_ptr())
    r.stream_wait_frame(ag.cuda_stream)
    with torch.cuda.stream(ag):
        keep[n].copy_(shard8[b] if present else shard[b], non_blocking=True)
        consumed[b].record(ag)
r.synchronize(); torch.cuda.synchronize()
ref = keep[steps - 1]   # the first frames may have outgrown an overflow (self-healing capacities): compare with the last
bad = [n for n in range(steps) if not torch.equal(keep[n].view(torch.uint8), ref.view(torch.uint8))]
print(f'fif {fif} steps {steps} present {present}: {len(bad)} frames differ from the last frame', bad[:12])
for n in bad[:3]:
    d = (keep[n].view(torch.uint8) != ref.view(torch.uint8)).reshape(rows, W, -1).any(dim=2)
    ys, xs = torch.nonzero(d, as_tuple=True)
    print(f'  frame {n}: {int(d.sum())} pixels, rows {int(ys.min())}..{int(ys.max())} x {int(xs.min())}..{int(xs.max())}')
scene.close(); r.close()
