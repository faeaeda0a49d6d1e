#!/bin/bash
# diagnostic: FETCH_SIZE / WRITE_SIZE per launch of every kernel (separate passes), one workload
W=${1:-c3}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/traffic_$W
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --workload $W --steps 40 --warmup 5 --no-cpu-baseline"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d ${O}_fetch --output-format csv -- $B > /dev/null 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d ${O}_write --output-format csv -- $B > /dev/null 2>&1
cd $R && python3 tools/profile_summary.py hbm $W ${O}_fetch ${O}_write $O.json && python3 -c "
import json; d=json.load(open('$O.json'))
for k,v in d['kernels'].items(): print('%-14s read %8.1f MB  write %8.1f MB  total %8.1f MB' % (k, v['read_bytes_corrected']/1e6, v['write_bytes']/1e6, v['hbm_bytes_per_launch']/1e6))"
