#!/bin/bash
# quick timing of the C3 frame (diagnostic): prints frame ms, k_shade pipelined / alone, parity
set -e
mkdir -p gpurun_out/r2
timeout -k 10 300 python bench.py --steps ${1:-100} --warmup 10 --no-cpu-baseline > gpurun_out/r2/quick.json 2> gpurun_out/r2/quick.err || { tail -20 gpurun_out/r2/quick.err; exit 1; }
python - <<'PY'
import json
d = json.load(open("gpurun_out/r2/quick.json"))
r = d["roofline"]; o = r["one_frame_in_flight"]
print(f"frame {d['ms_per_step']*1e3:.1f} us  value {d['value']:.0f}  k_shade pipelined {r['avg_kernel_ms']*1e3:.1f} us  alone: shade {o['avg_kernel_ms']*1e3:.1f} geom {o['avg_geometry_ms']*1e3:.1f} raster {o['avg_raster_ms']*1e3:.1f} us  layout {d['config'].get('stream_layout')}")
PY
