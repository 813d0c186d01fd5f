#!/bin/bash
# micro-benchmark of the ring-kernel layers under each tuning build given as argument (names of libmrisr_<name>.so), two rounds
mkdir -p gpurun_out/r3
F=${FILTER:-down1.3,down2.3,up1.c0,up2.c0}
for i in 1 2; do
  for v in base "$@"; do
    lib=mri_superresolution_amd/libmrisr_$v.so; [ $v = base ] && lib=mri_superresolution_amd/libmrisr.so
    echo "== $v round $i"
    MRISR_LIB=$PWD/$lib timeout -k 10 120 python tools/conv_bench.py --kinds ${KINDS:-dgrad} --iters 20 --filter $F $EXTRA 2>/dev/null | grep -v "total us"
  done
done
