#!/bin/bash
# Tuning aid: libmrisr_${TUNE_NAME:-tune}.so = libmrisr.so with the MRISR_DEBUG ablation switches of the convolution kernels
# compiled in (-DMRISR_TUNING; forward/dgrad and wgrad).  Results of ablated runs are invalid by construction: timing only.
#   MRISR_LIB=$PWD/mri_superresolution_amd/libmrisr_tune.so MRISR_DEBUG=8 python tools/conv_bench.py --kinds wgrad
set -e
cd "$(dirname "$0")/.."
python -m mri_superresolution_amd.build
mkdir -p build/tune
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -DMRISR_TUNING $TUNE_FLAGS"
/opt/rocm/bin/hipcc $F -fno-slp-vectorize -c mri_superresolution_amd/csrc/conv_fwd.hip -o build/tune/conv_fwd.o &
/opt/rocm/bin/hipcc $F -c mri_superresolution_amd/csrc/conv_wgrad.hip -o build/tune/conv_wgrad.o &
wait
objs=$(ls build/mrisr/*.o | grep -v -e conv_fwd.o -e conv_wgrad.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o mri_superresolution_amd/libmrisr_${TUNE_NAME:-tune}.so build/tune/conv_fwd.o build/tune/conv_wgrad.o $objs
echo built mri_superresolution_amd/libmrisr_${TUNE_NAME:-tune}.so
