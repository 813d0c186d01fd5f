#!/bin/bash
# weight-gradient CU share at the C5 shapes (f=128, depth 5, 512^2, fp16, batch 8)
C5="--base-filters 128 --depth 5 --size 512 --batch 8 --dtype fp16 --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-timer --no-forward-metric"
for r in 1 2; do for c in ${CUS_LIST:-96 104 112}; do
  v=$(MRISR_WGRAD_CUS=$c timeout -k 10 200 python bench.py $C5 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'])")
  echo "round $r wgrad_cus $c: $v"
done; done
