#!/bin/bash
# Tuning aid: libmrisr_${PROF_NAME:-prof}.so = libmrisr.so with the forward kernel's phase stamps compiled in
# (-DMRISR_PHASE_TIMING).  Use:  MRISR_LIB=$PWD/mri_superresolution_amd/libmrisr_${PROF_NAME:-prof}.so python tools/conv_bench.py --kinds fwd
set -e
cd "$(dirname "$0")/.."
python -m mri_superresolution_amd.build
mkdir -p build/prof
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -DMRISR_PHASE_TIMING -DMRISR_TUNING -fno-slp-vectorize $PROF_FLAGS \
    -c mri_superresolution_amd/csrc/conv_fwd.hip -o build/prof/conv_fwd.o
objs=$(ls build/mrisr/*.o | grep -v conv_fwd.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o mri_superresolution_amd/libmrisr_${PROF_NAME:-prof}.so build/prof/conv_fwd.o $objs
echo built mri_superresolution_amd/libmrisr_${PROF_NAME:-prof}.so
