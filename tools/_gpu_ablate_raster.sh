#!/bin/bash
# diagnostic (tools/_keep/ablate.so, -DBB_ABLATE): k_raster alone with a class of entries switched off
for w in c2 c3; do for a in 0 256 512 1024 4 1792 1796; do
  python tools/_gpu_variants.py --workload $w --opt ablate=$a tools/_keep/ablate.so | sed "s/^/$w ablate=$a /"
done; done
