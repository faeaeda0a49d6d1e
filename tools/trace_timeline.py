"""Per-stream timeline of a rocprofv3 kernel trace of bench.py: where a frame slot's stream spends its period.

usage: python tools/trace_timeline.py <dir with *_kernel_trace.csv> [first_frame last_frame]
For every hardware queue that carries frames: the kernels of each frame (k_geometry ... k_shade), their durations, the
idle gaps in front of each kernel, and the gap between a frame's last kernel and the next frame of the same queue."""
import csv
import glob
import sys
from collections import defaultdict


def short(name):
    for k in ("k_geometry", "k_raster", "k_shade_items", "k_shade_overlay", "k_shade", "k_present", "k_deferred_background",
              "k_pack", "k_unpack", "k_push"):
        if k in name:
            if k == "k_shade" and ("true>" in name.replace(" ", "") and name.replace(" ", "").endswith("true>(") is False):
                pass
            return k
    return None


def main():
    d = sys.argv[1]
    rows = []
    for f in glob.glob(d + "/**/*_kernel_trace.csv", recursive=True):
        with open(f) as fh:
            r = list(csv.DictReader(fh))
        if len(r) > len(rows):
            rows = r
    ks = []
    for r in rows:
        n = r["Kernel_Name"]
        s = short(n)
        if s is None:
            s = "copy" if "copyBuffer" in n else None
        if s is None:
            continue
        if s == "k_shade" and int(r["Grid_Size_X"]) == 32 * int(r["Workgroup_Size_X"]):
            s = "k_shade_tail"
        ks.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"]), s))
    ks.sort()
    byq = defaultdict(list)
    for k in ks:
        byq[k[2]].append(k)
    lo = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    hi = int(sys.argv[3]) if len(sys.argv) > 3 else 160
    # frames: sequences starting at k_geometry on a queue
    frames = []
    for q, lst in byq.items():
        cur = None
        for k in lst:
            if k[3] == "k_geometry":
                if cur:
                    frames.append(cur)
                cur = {"q": q, "k": []}
            if cur is not None and k[3] != "copy":
                cur["k"].append(k)
        if cur:
            frames.append(cur)
    frames.sort(key=lambda f: f["k"][0][0])
    sel = frames[lo:hi]
    if not sel:
        print("no frames", len(frames))
        return
    t0 = sel[0]["k"][0][0]
    period = (sel[-1]["k"][0][0] - t0) / 1e3 / (len(sel) - 1)
    print(f"# frames {lo}..{hi} of {len(frames)}; mean start-to-start {period:.1f} us; queues {sorted(set(f['q'] for f in sel))}")
    names = ["k_geometry", "k_raster", "k_shade_items", "k_shade_tail", "k_shade"]
    dur = defaultdict(list)
    gap = defaultdict(list)
    last_end = {}
    inter = []
    for f in frames[max(0, lo - 8):hi]:
        q = f["q"]
        prev_end = last_end.get(q)
        for k in f["k"]:
            if prev_end is not None and f in sel:
                (gap if k[3] != "k_geometry" else gap)[k[3]].append((k[0] - prev_end) / 1e3)
            if f in sel:
                dur[k[3]].append((k[1] - k[0]) / 1e3)
            prev_end = k[1]
        last_end[q] = prev_end
    tot = 0.0
    for n in names + [x for x in dur if x not in names]:
        if n in dur:
            g = gap.get(n, [0])
            md, mg = sum(dur[n]) / len(dur[n]), sum(g) / max(1, len(g))
            tot += md + mg
            print(f"{n:16s} n={len(dur[n]):4d}  idle in front {mg:8.2f} us   duration {md:8.2f} us")
    print(f"sum (one slot's period if its stream were never idle otherwise) {tot:.1f} us; frames in flight {tot / period:.2f}")
    # a few frames verbatim
    for f in sel[:4]:
        print("frame on queue", f["q"], " ".join(f"{k[3]}[{(k[0]-t0)/1e3:.1f}..{(k[1]-t0)/1e3:.1f}]" for k in f["k"]))


if __name__ == "__main__":
    main()


def concurrency(d, lo_frac=0.3, hi_frac=0.7):
    """share of the time by (geometry, raster, shade) kernels running at once, over the middle of the trace"""
    rows = []
    for f in glob.glob(d + "/**/*_kernel_trace.csv", recursive=True):
        with open(f) as fh:
            r = list(csv.DictReader(fh))
        if len(r) > len(rows):
            rows = r
    ev = []
    for r in rows:
        s = short(r["Kernel_Name"])
        if s not in ("k_geometry", "k_raster", "k_shade", "k_shade_items"):
            continue
        if s == "k_shade" and int(r["Grid_Size_X"]) == 32 * int(r["Workgroup_Size_X"]):
            s = "tail"
        ev.append((int(r["Start_Timestamp"]), 1, s))
        ev.append((int(r["End_Timestamp"]), -1, s))
    ev.sort()
    t_lo = ev[0][0] + (ev[-1][0] - ev[0][0]) * lo_frac
    t_hi = ev[0][0] + (ev[-1][0] - ev[0][0]) * hi_frac
    cnt = defaultdict(int)
    state = defaultdict(float)
    prev = None
    for t, dlt, s in ev:
        if prev is not None and t_lo <= prev and t <= t_hi:
            state[(cnt["k_geometry"], cnt["k_raster"], cnt["k_shade"])] += t - prev
        cnt[s] += dlt
        prev = t
    tot = sum(state.values())
    print("# (geometry, raster, shade) kernels running at once: share of the time")
    for k, v in sorted(state.items(), key=lambda kv: -kv[1])[:16]:
        print(f"  G{k[0]} R{k[1]} S{k[2]}  {100 * v / tot:5.1f} %")


if __name__ == "__main__" and len(sys.argv) > 1:
    concurrency(sys.argv[1])
