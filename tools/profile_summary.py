"""Turn rocprofv3 output directories into the small summaries committed under profiles/.

  python tools/profile_summary.py stats <rocprof_dir> <out.csv>
      copy the --kernel-trace --stats per-kernel table (kernel names shortened)
  python tools/profile_summary.py hbm <workload> <fetch_dir> <write_dir> <out.json>
      FETCH_SIZE and WRITE_SIZE (separate --pmc passes) -> HBM bytes per launch and kernel.  rocprofv3 reports both
      in KiB; on gfx950 FETCH_SIZE counts 128-byte requests as 64 bytes and is doubled, WRITE_SIZE is exact
      (MI355X_MICROARCH.md, section HBM).
  python tools/profile_summary.py phases <rocprof_dir> <bench.json> <out.txt>
      mean kernel duration from the kernel trace, split into bench.py's phases: the timed region (the launches after
      roofline.frames_before_timed_region of each kernel) and the trailing one-frame-in-flight pass
  python tools/profile_summary.py counters_json <workload> <out.json> <dir>...
      the per-launch counter means as JSON, stamped with the hash of the kernel sources
  python tools/profile_summary.py counters <out.txt> <dir>...
      mean per launch of every counter in the given --pmc passes
"""
import collections, csv, glob, json, re, sys


def short(name):
    """bbr::k_shade<32, 32, false, false, true>(...) -> k_shade_tail (the TAIL instantiation: the few workgroups that cover
    what the main launch's size estimate missed), every other kernel -> its plain name"""
    m = re.search(r"(k_[a-z_0-9]+)(<[^>]*>)?", name)
    if not m:
        return name.split("(")[0]
    base = m.group(1)
    if base == "k_shade" and m.group(2):
        args = [a.strip() for a in m.group(2)[1:-1].split(",")]
        if len(args) >= 5 and args[4] in ("true", "1"):
            base = "k_shade_tail"
    return base


def source_hash():
    """what a committed counter summary was measured on (bibim_renderer_amd/build_id.py)"""
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bibim_renderer_amd.build_id import kernel_source_sha256
    return kernel_source_sha256()


def counter_means(root):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            acc[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
    out = {}
    for k, d in acc.items():
        out[k] = {}
        for c, v in d.items():
            v = v[len(v) // 4:]  # skip warm-up launches
            out[k][c] = (sum(v) / len(v), len(v))
    return out


mode = sys.argv[1]
if mode == "stats":
    src = glob.glob(sys.argv[2] + "/**/*kernel_stats.csv", recursive=True)[0]
    rows = list(csv.reader(open(src)))
    with open(sys.argv[3], "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(rows[0])
        for r in rows[1:]:
            w.writerow([short(r[0])] + r[1:])
elif mode == "hbm":
    workload, fetch, write, out = sys.argv[2:6]
    fm, wm = counter_means(fetch), counter_means(write)
    kernels = {}
    for k in fm:
        if not k.startswith("k_") or "FETCH_SIZE" not in fm[k] or "WRITE_SIZE" not in wm.get(k, {}):
            continue
        f_kib, n = fm[k]["FETCH_SIZE"]
        w_kib, _ = wm[k]["WRITE_SIZE"]
        kernels[k] = {"launches_averaged": n, "FETCH_SIZE_KiB": round(f_kib, 1), "WRITE_SIZE_KiB": round(w_kib, 1),
                      "read_bytes_corrected": int(f_kib * 1024 * 2), "write_bytes": int(w_kib * 1024),
                      "hbm_bytes_per_launch": int(f_kib * 1024 * 2 + w_kib * 1024)}
    json.dump({"workload": workload, "correction": "FETCH_SIZE x2 (gfx950), WRITE_SIZE x1; KiB -> bytes",
               "kernel_source_sha256": source_hash(), "kernels": kernels}, open(out, "w"), indent=1)
elif mode == "phases":
    # python tools/profile_summary.py phases <rocprof_dir> <bench.json> <out.txt>: the bench line says how many frames it
    # submitted before the timed region and how many steps it timed
    src = glob.glob(sys.argv[2] + "/**/*kernel_trace.csv", recursive=True)[0]
    bench = json.load(open(sys.argv[3]))
    skip, steps = int(bench["roofline"]["frames_before_timed_region"]), int(bench["steps"])
    per = collections.defaultdict(list)
    for row in csv.DictReader(open(src)):
        per[short(row["Kernel_Name"])].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    with open(sys.argv[4], "w") as f:
        f.write(f"# {skip} frames before the timed region, {steps} timed steps ({bench['config']['workload'].split(':')[0]}, "
                f"{bench['roofline']['frames_in_flight']} frames in flight); durations from the kernel trace\n")
        f.write(f"# kernel_source_sha256 {source_hash()}\n")   # (bench.py quotes these durations only for the kernels of its own tree)
        for k, v in sorted(per.items()):
            if not k.startswith("k_"):
                continue
            if len(v) < skip + steps:  # not a per-frame kernel (k_present: only the next-row measurement launches it)
                f.write(f"{k:14s} launches {len(v):4d}  all: mean={sum(v) / len(v) / 1e3:8.2f} us\n")
                continue
            timed, tail = v[skip:skip + steps], v[skip + steps + 5:skip + steps + 55]
            f.write(f"{k:14s} launches {len(v):4d}  timed region: n={len(timed)} mean={sum(timed) / max(len(timed), 1) / 1e3:8.2f} us"
                    f"   one frame in flight: n={len(tail)} mean={sum(tail) / max(len(tail), 1) / 1e3:8.2f} us\n")
elif mode == "counters_json":
    # python tools/profile_summary.py counters_json <workload> <out.json> <dir>...: the same means, machine readable, with
    # the hash of the sources they were measured on (bench.py reads roofline.valu from it)
    workload, out = sys.argv[2:4]
    kernels = collections.defaultdict(dict)
    for d in sys.argv[4:]:
        for k, cs in counter_means(d).items():
            if k.startswith("k_"):
                for c, (mean, n) in cs.items():
                    kernels[k][c] = round(mean, 1)
                    kernels[k]["launches_averaged"] = n
    json.dump({"workload": workload, "kernel_source_sha256": source_hash(), "kernels": kernels}, open(out, "w"), indent=1, sort_keys=True)
elif mode == "counters":
    with open(sys.argv[2], "w") as f:
        for d in sys.argv[3:]:
            for k, cs in sorted(counter_means(d).items()):
                if not k.startswith("k_"):
                    continue
                for c, (mean, n) in sorted(cs.items()):
                    f.write(f"{k:14s} {c:28s} n={n:4d} mean={mean:16.1f}\n")
