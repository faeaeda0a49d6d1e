#!/bin/bash
# diagnostic (GPU box): shader clock and socket power while the C3 bench runs (sampled with rocm-smi beside it)
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
( for i in $(seq 1 60); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Average Graphics Package Power|Current Socket Graphics Package Power" | tr '\n' ' '; echo; sleep 0.25; done ) > gpurun_out/clocks.txt &
S=$!
python3 bench.py --no-cpu-baseline --steps 60000 --warmup 20 --no-timing-events > gpurun_out/clocks_bench.json 2>/dev/null
kill $S 2>/dev/null
sort gpurun_out/clocks.txt | uniq -c | sort -rn | head -12
python3 -c "import json; d=json.load(open('gpurun_out/clocks_bench.json')); print('ms/step', d['ms_per_step'])"
