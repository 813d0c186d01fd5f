#!/bin/bash
# A/B of environment settings on ONE box (alternating runs): tools/ab_env.sh <rounds> "<VAR=a>" "<VAR=b>" ... [-- bench args]
R=$1; shift
SETS=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do SETS+=("$1"); shift; done
[ "$1" == "--" ] && shift
for r in $(seq $R); do
  for S in "${SETS[@]}"; do
    env $S timeout -k 10 150 python bench.py --no-cpu-baseline --no-kernel-timer --no-forward-metric --steps 30 "$@" 2>/dev/null | tail -1 \
      | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$S', j['value'], j['ms_per_step'])"
  done
done
