#!/bin/bash
# Tuning aid: libmrisr_<name>.so = libmrisr.so with conv_fwd.hip recompiled with extra flags, for A/B runs on ONE box
# (box-to-box spread is +-1.5 %):   tools/build_variant.sh noprio -DMRISR_NO_STATIC_PRIO
#   MRISR_LIB=$PWD/mri_superresolution_amd/libmrisr_noprio.so python bench.py ...
set -e
cd "$(dirname "$0")/.."
name=$1; shift
python -m mri_superresolution_amd.build
mkdir -p build/$name
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -fno-slp-vectorize "$@" \
    -c mri_superresolution_amd/csrc/conv_fwd.hip -o build/$name/conv_fwd.o
objs=$(ls build/mrisr/*.o | grep -v conv_fwd.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o mri_superresolution_amd/libmrisr_$name.so build/$name/conv_fwd.o $objs
echo built mri_superresolution_amd/libmrisr_$name.so
