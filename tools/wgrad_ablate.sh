#!/bin/bash
export MRISR_LIB=$PWD/mri_superresolution_amd/libmrisr_tune.so
for d in 0 8 2 4 6 1 9 14 10; do
  echo "== MRISR_DEBUG=$d"
  MRISR_DEBUG=$d timeout -k 10 120 python tools/conv_bench.py --kinds wgrad --filter "down1.3,down2.3,down3.3,up2.c0,inc.3" --iters 20 2>/dev/null | grep -v "^total"
done
