#!/bin/bash
# Per-kernel A/B inside the real training step on ONE box: tools/ab_kernels.sh <libA.so> <libB.so>
for L in "$1" "$2" "$1" "$2"; do
  MRISR_LIB=$PWD/mri_superresolution_amd/$L timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 2>/dev/null | tail -1 \
    | python -c "
import sys,json
j=json.loads(sys.stdin.read()); print('== $L', j['value'], 'slices/s')
for k,v in j['kernels'].items(): print(f\"   {v['us_per_launch']:8.1f} us x {v['launches']:4d}  {k}\")"
done
