#!/bin/bash
# round 4: rehearsals of the supervisor under a launcher, then what the timing events cost the driver's 20-step line
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4; mkdir -p $O; cd $R
timeout -k 10 1000 python3 -m pytest tests/test_gpu_bench_rehearsal.py -m gpu -x -q > $O/exp5_pytest.txt 2>&1; echo "pytest rc=$?" >> $O/exp5_pytest.txt
tail -8 $O/exp5_pytest.txt
for v in "" "--event-stride 10" "--no-timing-events" "" "--event-stride 10" "--no-timing-events"; do
  timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline $v 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['ms_per_step'], (d['roofline'] or {}).get('avg_kernel_ms'))"
done
