#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4; mkdir -p $O; cd $R
for p in 0 1 2 3 0 1; do
  echo "BBR_STREAM_PRIORITIES=$p"
  BBR_STREAM_PRIORITIES=$p python3 tools/_gpu_rate.py --reps 3 c3 c5 c2
done > $O/exp6_prio.txt 2>&1
cat $O/exp6_prio.txt
