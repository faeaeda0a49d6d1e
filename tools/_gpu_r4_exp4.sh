#!/bin/bash
# round 4, experiment 4: k_shade in two load groups at 58 registers (MIXED = false instantiation): whole GPU suite, then rates
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4; mkdir -p $O; cd $R
timeout -k 10 800 python3 -m pytest tests -m gpu -x -q > $O/exp4_pytest.txt 2>&1; echo "pytest rc=$?" >> $O/exp4_pytest.txt
tail -5 $O/exp4_pytest.txt
python3 tools/_gpu_rate.py --reps 3 c3 c5 c2 c3 c5 > $O/exp4_rate.txt 2>&1
cat $O/exp4_rate.txt
for W in c3 c5 c2; do timeout -k 10 200 python3 tools/_gpu_variants.py --workload $W bibim_renderer_amd/libbibim_hip.so >> $O/exp4_rate.txt 2>&1; done
tail -3 $O/exp4_rate.txt
