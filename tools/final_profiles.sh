#!/bin/bash
# Run on the GPU box from the repo root (via gpurun), LAST, on the final kernel sources: the round's rocprofv3 evidence for
# C3 / C2 / C5 and then the bench lines, which quote roofline.traffic / roofline.valu from the counter summaries just taken
# (so those are copied into profiles/ of the box's copy first).  usage: tools/final_profiles.sh r03 [skip-profiles]
tag=$1
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
mkdir -p $O
cd $R
if [ -z "$2" ]; then
  for W in c3 c2 c5; do timeout -k 10 500 tools/profile_round.sh $tag $W > $O/${tag}_profile_$W.log 2>&1 || echo "profile $W failed"; done
fi
cp $O/${tag}_pmc_*.json $O/${tag}_c2_pmc_*.json $O/${tag}_c5_pmc_*.json $O/${tag}_*kernel_phases.txt profiles/ 2>/dev/null
python3 bench.py > $O/${tag}_bench.json 2> $O/${tag}_bench.err
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/${tag}_bench_driver_style.json 2>> $O/${tag}_bench.err
python3 bench.py --workload c2 > $O/${tag}_c2_bench.json 2>> $O/${tag}_bench.err
python3 bench.py --workload c5 --steps 60 > $O/${tag}_c5_bench.json 2>> $O/${tag}_bench.err
python3 tools/shade_issue_floor.py $O/${tag}_pmc_sq.json $(python3 -c "import json;print(json.loads(open('$O/${tag}_bench.json').read().strip().splitlines()[-1])['roofline']['one_frame_in_flight']['avg_kernel_ms']*1e3)") > $O/${tag}_k_shade_issue_floor.txt 2>&1
for f in $O/${tag}_bench.json $O/${tag}_bench_driver_style.json $O/${tag}_c2_bench.json $O/${tag}_c5_bench.json; do python3 -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1]); r=d['roofline']
print('$f'.split('/')[-1], d['ms_per_step'], d['value'], 'traffic', r['traffic'], r['traffic_source'], 'valu', (r.get('valu') or {}).get('frac'), r['limiter'])"; done
