mkdir -p gpurun_out/r3
C5="--base-filters 128 --depth 5 --size 512 --batch 8 --dtype fp16 --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timer"
for i in 1 2; do
for v in base NO_RING NO_ONEPASS UP_FUSED; do
  unset MRISR_NO_RING MRISR_NO_ONEPASS MRISR_UP_FUSED
  [ $v != base ] && export MRISR_$v=1
  python bench.py $C5 2>/dev/null | tail -1 | python -c "
import sys,json
j=json.loads(sys.stdin.read()); print('$v', j['value'], 'slices/s', j['ms_per_step'], 'ms; fwd', j['forward']['slices_per_s'])"
done
done
