#!/bin/bash
# diagnostic (needs tools/_keep/ablate.so built with -DBB_ABLATE): k_shade alone with parts of it switched off
for a in 0 4096 8192 2048 8 2056 10240 10248; do
  echo "ablate=$a"; python tools/_gpu_variants.py --opt ablate=$a tools/_keep/ablate.so
done
