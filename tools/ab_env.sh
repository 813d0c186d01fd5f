#!/bin/bash
# In-step A/B of a host-side tuning switch on ONE box: tools/ab_env.sh MRISR_NO_ONEPASS  -> bench.py alternating VAR unset / VAR=1
var=$1
for i in 1 2 3; do
  for v in 0 1; do
    if [ $v = 1 ]; then export $var=1; else unset $var; fi
    timeout -k 10 200 python bench.py --no-cpu-baseline --no-kernel-timer --steps 30 2>/dev/null | tail -1 | python -c "
import sys,json
j=json.loads(sys.stdin.read()); print('round $i $var=$v', j['value'], 'slices/s', j['ms_per_step'], 'ms; fwd', j['forward']['slices_per_s'], j['forward']['ms_per_batch'])"
  done
done
