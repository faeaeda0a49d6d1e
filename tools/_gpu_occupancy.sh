#!/bin/bash
# diagnostic: average resident waves per kernel while three frames are in flight (SQ_LEVEL_WAVES / SQ_BUSY_CYCLES ...)
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/occ
cd /tmp && export TMPDIR=/tmp
rm -rf $O
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_LEVEL_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE -d $O --output-format csv -- python3 $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
cd $R && python3 tools/profile_summary.py counters $O.txt $O && cat $O.txt
