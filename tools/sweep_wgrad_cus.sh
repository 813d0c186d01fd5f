#!/bin/bash
# step time against the CU split between the gradient chain (main stream) and the weight gradients (second stream)
for r in 1 2; do for c in ${CUS_LIST:-96 112 128 144 160}; do
  v=$(MRISR_WGRAD_CUS=$c timeout -k 10 200 python bench.py --no-cpu-baseline --no-kernel-timer --no-forward-metric --steps 30 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'])")
  echo "round $r wgrad_cus $c: $v"
done; done
