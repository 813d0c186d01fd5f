#!/bin/bash
# In-step A/B on ONE box: bench.py alternating between libmrisr.so and the tuning builds given as arguments (names of
# libmrisr_<name>.so), two rounds; prints slices/s, forward slices/s and the per-kernel table of the instrumented pass.
mkdir -p gpurun_out/r3
for i in 1 2; do
  for v in base "$@"; do
    lib=mri_superresolution_amd/libmrisr_$v.so; [ $v = base ] && lib=mri_superresolution_amd/libmrisr.so
    MRISR_LIB=$PWD/$lib timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 $BENCH_ARGS 2>/dev/null | tail -1 > gpurun_out/r3/abl_${v}_$i.json
  done
done
python - "$@" <<'PY'
import json, sys
for i in (1, 2):
    for n in ["base"] + sys.argv[1:]:
        j = json.load(open(f"gpurun_out/r3/abl_{n}_{i}.json"))
        print("==", n, i, j["value"], "slices/s", j["ms_per_step"], "ms; fwd", j.get("forward", {}).get("slices_per_s"))
        for k, v in list(j.get("kernels", {}).items())[:int(__import__("os").environ.get("NK", "8"))]:
            print(f"   {v['us_per_launch']:8.1f} us x {v['launches']:4d} {v['tflops']:7.1f} TF {k}")
PY
