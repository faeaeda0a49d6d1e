#!/bin/bash
# diagnostic (GPU box): A/B of several builds of the library.  usage: tools/_gpu_ab.sh <outdir> lib1.so [lib2.so ...]
# per build: every kernel alone (C3, C2), then the pipelined bench lines (C3 200 steps, C3 driver-style, C2)
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/$1; shift
mkdir -p $O
cd $R
for W in c3 c2; do
  timeout -k 10 300 python3 tools/_gpu_variants.py --workload $W "$@" > $O/alone_$W.txt 2>&1 || exit 1
done
for lib in "$@"; do
  n=$(basename $lib .so)
  BBR_LIB=$R/$lib timeout -k 10 200 python3 bench.py --no-cpu-baseline > $O/bench_c3_$n.json 2> $O/bench_c3_$n.err || exit 1
  BBR_LIB=$R/$lib timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/bench_c3drv_$n.json 2>> $O/bench_c3_$n.err || exit 1
  BBR_LIB=$R/$lib timeout -k 10 200 python3 bench.py --no-cpu-baseline --workload c2 > $O/bench_c2_$n.json 2>> $O/bench_c3_$n.err || exit 1
done
cat $O/alone_c3.txt $O/alone_c2.txt
python3 - <<PY
import json, glob, os
for f in sorted(glob.glob("$O/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d["roofline"]; o = r.get("one_frame_in_flight", {})
        print(f"{os.path.basename(f):34s} ms/step {d['ms_per_step']:.5f}  value {d['value']:.0f}  k_shade ev {r['avg_kernel_ms']:.4f}  alone: shade {o.get('avg_kernel_ms')} raster {o.get('avg_raster_ms')} geom {o.get('avg_geometry_ms')} latency {o.get('avg_device_frame_latency_ms')}  fif2 {r.get('frames_in_flight_2', {}).get('ms_per_step')}")
    except Exception as e:
        print(f, "unreadable", e)
PY
