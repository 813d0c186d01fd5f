mkdir -p gpurun_out/r3
for i in 1 2; do
  timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 2>/dev/null | tail -1 > gpurun_out/r3/ab_ring_$i.json
  MRISR_NO_RING=1 timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 2>/dev/null | tail -1 > gpurun_out/r3/ab_noring_$i.json
done
python - <<'PY'
import json
for n in ("ring_1","noring_1","ring_2","noring_2"):
    j=json.load(open(f"gpurun_out/r3/ab_{n}.json"))
    print("==",n,j["value"],"slices/s",j["ms_per_step"],"ms; fwd",j["forward"]["slices_per_s"])
    for k,v in j["kernels"].items(): print(f"   {v['us_per_launch']:8.1f} us x {v['launches']:4d} {v['tflops']:7.1f} TF {k}")
PY
