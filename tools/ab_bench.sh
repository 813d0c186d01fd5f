#!/bin/bash
# A/B of two library builds on ONE box (alternating runs): tools/ab_bench.sh <libA.so> <libB.so> [rounds] [bench args...]
A=$1; B=$2; R=${3:-3}; shift 3
for r in $(seq $R); do
  for L in "$A" "$B"; do
    MRISR_LIB=$PWD/mri_superresolution_amd/$L timeout -k 10 120 python bench.py --no-cpu-baseline --no-kernel-timer --steps 30 "$@" 2>/dev/null | tail -1 \
      | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$L', j['value'], j['ms_per_step'])"
  done
done
