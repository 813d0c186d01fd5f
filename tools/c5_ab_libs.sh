#!/bin/bash
# In-step A/B of library builds on BASELINE config 5 (f = 128, depth 5, 512^2, batch 8, fp16) on ONE box:
#   tools/c5_ab_libs.sh <name> ...   (names of libmrisr_<name>.so; "base" = libmrisr.so), two alternating rounds
C5="--base-filters 128 --depth 5 --size 512 --batch 8 --dtype fp16 --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timer"
for i in 1 2; do
for v in base "$@"; do
  lib=mri_superresolution_amd/libmrisr_$v.so; [ $v = base ] && lib=mri_superresolution_amd/libmrisr.so
  MRISR_LIB=$PWD/$lib timeout -k 10 300 python bench.py $C5 2>/dev/null | tail -1 | python -c "
import sys,json
j=json.loads(sys.stdin.read()); print('$v', j['value'], 'slices/s', j['ms_per_step'], 'ms; fwd', j['forward']['slices_per_s'])"
done
done
