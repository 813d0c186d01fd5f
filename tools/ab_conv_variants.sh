#!/bin/bash
# Per-layer A/B on ONE box: tools/conv_bench.py with libmrisr.so and the tuning builds given as arguments (names of
# libmrisr_<name>.so).  FILTER / KINDS select layers and launch kinds.
mkdir -p gpurun_out/r3
for v in base "$@"; do
  lib=mri_superresolution_amd/libmrisr_$v.so; [ $v = base ] && lib=mri_superresolution_amd/libmrisr.so
  echo "== $v"
  MRISR_LIB=$PWD/$lib timeout -k 10 200 python tools/conv_bench.py --iters 20 --kinds ${KINDS:-fwd} --filter "${FILTER:-down1.3,up1.c0}" 2>&1 | grep "k3" | cut -c1-170
done
