#!/bin/bash
# round 4, experiment 2: kernarg preload + one-round-trip light tiles: GPU suite, then rates
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4; mkdir -p $O; cd $R
timeout -k 10 700 python3 -m pytest tests -m gpu -x -q > $O/exp2_pytest.txt 2>&1; echo "pytest rc=$?" >> $O/exp2_pytest.txt
tail -5 $O/exp2_pytest.txt
python3 tools/_gpu_rate.py --reps 3 c3 c5 c2 c3 > $O/exp2_rate.txt 2>&1
cat $O/exp2_rate.txt
timeout -k 10 200 python3 tools/_gpu_variants.py --workload c3 bibim_renderer_amd/libbibim_hip.so >> $O/exp2_rate.txt 2>&1
tail -2 $O/exp2_rate.txt
