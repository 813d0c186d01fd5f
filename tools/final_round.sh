#!/bin/bash
# Round-end evidence on ONE GPU box (inside gpurun): profiles (tools/refresh_profiles.sh), the whole -m gpu suite, smoke(), the bench
# lines of the secondary configurations and the per-layer convolution table, all into gpurun_out/<dir>/ for copying into profiles/.
#   bash tools/final_round.sh r03 fin
tag=${1:-r03}; out=gpurun_out/${2:-fin}
mkdir -p $out
timeout -k 10 400 bash tools/refresh_profiles.sh $tag 2>/dev/null
timeout -k 10 700 python -m pytest tests -q -m gpu -x > $out/gpu_tests.log 2>&1; tail -2 $out/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1 && tail -1 $out/smoke.log || exit 1
python bench.py --dtype fp16 --no-cpu-baseline > $out/${tag}_bench_c2_fp16.log 2>&1 || exit 1
python bench.py --perceptual-weight 0.1 --no-cpu-baseline > $out/${tag}_bench_c3_perceptual.log 2>&1 || exit 1
python bench.py --forward-only --no-cpu-baseline > $out/${tag}_bench_forward_only.log 2>&1 || exit 1
MRISR_FORCE_DP=1 python bench.py --no-cpu-baseline > $out/${tag}_bench_force_dp.log 2>&1 || exit 1
python bench.py --base-filters 128 --depth 5 --size 512 --batch 8 --dtype fp16 --steps 10 --warmup 3 --no-cpu-baseline > $out/${tag}_bench_c5_depth5_f128_512_fp16.log 2>&1 || exit 1
python tools/conv_bench.py --iters 20 > $out/conv_layers.log 2>&1 || exit 1
python tools/conv_bench.py --iters 20 --no-stats --filter up1.up,up2.up,up3.up --kinds fwd > $out/conv_layers_nostats.log 2>&1
for f in $out/${tag}_bench_*.log; do tail -1 $f | python tools/bench_summary.py /dev/stdin 2>/dev/null | sed -n 1p; done
