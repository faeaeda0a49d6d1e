"""How many texel bytes a frame NEEDS, against what k_shade fetches.

No gfx950 counter separates DRAM reads from Infinity-Cache hits (rocprofv3 -L: SQ / SPI / TCC / TCP / TA / TD / CPC blocks
only; FETCH_SIZE is the L2's memory-side request count, Infinity-Cache hits included), so the DRAM share is bounded from
the other side: the CPU oracle renders the winning fragment's vUV of every pixel (BBO_FLAG_OUTPUT_UV), this script turns it
into the four bilinear taps exactly as the kernel addresses them (9-byte packed texels, 4 x 4 block-linear,
csrc/bb_kernels.hip.h bilinear_taps<true>) and counts the DISTINCT 128-byte lines and 64-byte half lines of the packed
material the frame touches -- the compulsory texel traffic: every byte beyond it that the counters report is the same line
fetched again (another XCD's L2, or evicted in between), which the 256 MB Infinity Cache serves as long as the lines fit.

usage: python tools/texel_lines.py [c3|c2|c5]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bibim_renderer_amd import configs, textures
from oracle import bbo, scenes

name = sys.argv[1] if len(sys.argv) > 1 else "c3"
cfg = configs.CONFIGS[name]
T = cfg.texture_size
sc = scenes.shaderball_scene(cfg, bbo.MaterialData(textures.make_material(64)))   # (the texels themselves do not matter here)
uv, n_shaded = bbo.render_bands(sc, flags=bbo.FLAG_OUTPUT_UV)
cov = uv[..., 3] == 1.0
u = uv[..., 0][cov].astype(np.float32); v = uv[..., 1][cov].astype(np.float32)
x = u * np.float32(T) - np.float32(0.5); y = v * np.float32(T) - np.float32(0.5)
ix = np.floor(x).astype(np.int64); iy = np.floor(y).astype(np.int64)
w4 = (T + 3) // 4
lines128, lines64, blocks = set(), set(), None
acc128, acc64, accb = [], [], []
for dx in (0, 1):
    for dy in (0, 1):
        xx = (ix + dx) & (T - 1); yy = (iy + dy) & (T - 1)
        rec = ((yy >> 2) * w4 + (xx >> 2)) * 16 + (yy & 3) * 4 + (xx & 3)
        off = rec * 9
        for b in (off, off + 11):   # a tap is one 12-byte load: first and last byte
            acc128.append(b >> 7); acc64.append(b >> 6)
        accb.append(rec >> 4)
u128 = np.unique(np.concatenate(acc128)); u64 = np.unique(np.concatenate(acc64)); ub = np.unique(np.concatenate(accb))
taps = 4 * u.size
print(f"{cfg.name}: {n_shaded} shaded pixels, {taps} taps of 12 bytes = {taps * 12 / 1e6:.1f} MB requested by the lanes; "
      f"algorithmic (SURVEY 8(d): 4 B x 5 maps per shaded pixel) {n_shaded * 20 / 1e6:.1f} MB")
print(f"  packed material {T}x{T}: {T * T * 9 / 1e6:.1f} MB = {T * T * 9 // 128} lines of 128 B")
print(f"  distinct 4x4 texel blocks touched: {ub.size} ({ub.size * 144 / 1e6:.1f} MB)")
print(f"  distinct 128-byte lines touched:   {u128.size} = {u128.size * 128 / 1e6:.1f} MB   <- compulsory texel reads of the frame")
print(f"  distinct 64-byte half lines:       {u64.size} = {u64.size * 64 / 1e6:.1f} MB")
