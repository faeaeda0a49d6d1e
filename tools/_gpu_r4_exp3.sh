#!/bin/bash
# round 4, experiment 3b: k_shade with its dependent loads in two groups, 7 / 6 waves per SIMD (no scratch)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4; mkdir -p $O; cd $R
V=tools/_keep/variants
python3 tools/_gpu_rate.py --reps 3 c3:$V/w7.so c3:$V/w6.so c5:$V/w7.so c5:$V/w6.so c2:$V/w7.so c3 > $O/exp3b_rate.txt 2>&1
cat $O/exp3b_rate.txt
for W in c3 c5; do timeout -k 10 200 python3 tools/_gpu_variants.py --workload $W $V/w7.so $V/w6.so >> $O/exp3b_rate.txt 2>&1; done
tail -4 $O/exp3b_rate.txt
