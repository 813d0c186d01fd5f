#!/bin/bash
# Tuning aid: libmrisr_<name>.so = libmrisr.so with ONE csrc file recompiled with extra flags, for A/B runs on one box:
#   tools/build_src_variant.sh norows conv_wgrad_rows.hip -DMRISR_NO_WGRAD_ROWS
#   MRISR_LIB=$PWD/mri_superresolution_amd/libmrisr_norows.so python tools/conv_bench.py --kinds wgrad
set -e
cd "$(dirname "$0")/.."
name=$1; src=$2; shift; shift
python -m mri_superresolution_amd.build > /dev/null
mkdir -p build/$name
extra=""; [ "$src" = conv_fwd.hip ] && extra="-fno-slp-vectorize"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result $extra "$@" \
    -c mri_superresolution_amd/csrc/$src -o build/$name/${src%.*}.o
objs=$(ls build/mrisr/*.o | grep -v "/${src%.*}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o mri_superresolution_amd/libmrisr_$name.so build/$name/${src%.*}.o $objs
echo built mri_superresolution_amd/libmrisr_$name.so
