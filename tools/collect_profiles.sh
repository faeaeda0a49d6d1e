#!/bin/bash
# copy the summaries tools/final_profiles.sh left in gpurun_out/ into profiles/ (what the judge reads)   usage: tools/collect_profiles.sh r03
tag=$1
cd "$(dirname "$0")/.."
for w in "" c2_ c5_; do
  for f in bench.json bench_kernel_phases.txt bench_kernel_stats.csv bench_under_rocprof.json pmc_hbm.json pmc_sq.json; do
    [ -f gpurun_out/${tag}_${w}$f ] && cp gpurun_out/${tag}_${w}$f profiles/${tag}_${w}$f
  done
done
for f in bench_driver_style.json pmc_sq.txt k_shade_issue_floor.txt band_kernels.txt; do [ -f gpurun_out/${tag}_$f ] && cp gpurun_out/${tag}_$f profiles/${tag}_$f; done
python3 -c "
import json
from bibim_renderer_amd.build_id import kernel_source_sha256 as k
for w in ('', 'c2_', 'c5_'):
    for f in ('pmc_hbm', 'pmc_sq'):
        d = json.load(open('profiles/${tag}_%s%s.json' % (w, f)))
        assert d['kernel_source_sha256'] == k(), ('stale summary', w, f)
print('profiles/${tag}_*: kernel sources', k()[:12])"
