#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4; mkdir -p $O; cd $R
V=tools/_keep/variants
python3 tools/_gpu_rate.py --reps 3 c3 c3:$V/bilin.so c5 c5:$V/bilin.so c3 c3:$V/bilin.so c5 c5:$V/bilin.so > $O/exp7_bilin.txt 2>&1
for W in c3 c5; do timeout -k 10 200 python3 tools/_gpu_variants.py --workload $W bibim_renderer_amd/libbibim_hip.so $V/bilin.so bibim_renderer_amd/libbibim_hip.so $V/bilin.so >> $O/exp7_bilin.txt 2>&1; done
cat $O/exp7_bilin.txt
