#!/bin/bash
# Tuning aid: libmrisr_<name>.so = libmrisr.so with csrc/conv_ring.hip recompiled with extra flags, for A/B runs on ONE box:
#   tools/build_ring_variant.sh nodma -DMRISR_RING_DBG=1
#   MRISR_LIB=$PWD/mri_superresolution_amd/libmrisr_nodma.so python tools/conv_bench.py --kinds dgrad
# MRISR_RING_DBG bits (results invalid by construction, timing only): 1 no DMA issue, 2 no MFMA, 4 no epilogue
set -e
cd "$(dirname "$0")/.."
name=$1; shift
python -m mri_superresolution_amd.build
mkdir -p build/$name
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result "$@" \
    -c mri_superresolution_amd/csrc/conv_ring.hip -o build/$name/conv_ring.o
objs=$(ls build/mrisr/*.o | grep -v conv_ring.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o mri_superresolution_amd/libmrisr_$name.so build/$name/conv_ring.o $objs
echo built mri_superresolution_amd/libmrisr_$name.so
