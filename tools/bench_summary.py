#!/usr/bin/env python3
"""Prints the headline numbers and the per-kernel table of a bench.py JSON line (last line of the given file)."""
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
fw = j.get("forward") or {}
print(f"{j['dtype']} {j['value']} slices/s  {j['ms_per_step']} ms/step  fwd {fw.get('slices_per_s')} slices/s ({fw.get('frac_of_mfma_peak')})  "
      f"roofline {j.get('roofline', {}).get('kernel')} {j.get('roofline', {}).get('frac')}  loss {j.get('loss')}")
for k, v in (j.get("kernels") or {}).items():
    print(f"  {v['us_per_launch']:8.1f} us x {v['launches']:4d} {v['tflops']:7.1f} TF {v['share_of_step']*100:5.1f}%  {k}")
