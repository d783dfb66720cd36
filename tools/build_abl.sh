#!/bin/bash
# Builds tools/abl/libsamaudio_hip_abl.so: the product library with gemm8.hip compiled with -DSAMAUDIO_GEMM8_ABL (ablation /
# timestamp instantiations of the 8-phase kernel, selected by debug flag 25).  Timing experiments only - use it through
# SAMAUDIO_LIB_AB=tools/abl/libsamaudio_hip_abl.so (sam_audio_amd/hip.py).  Needs sam_audio_amd/csrc/build.sh to have run.
set -e
cd "$(dirname "$0")/../sam_audio_amd/csrc"
mkdir -p ../../tools/abl
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -mllvm -pragma-unroll-threshold=262144 -Wno-inline-asm \
  -DSAMAUDIO_GEMM8_ABL -c gemm8.hip -o ../../tools/abl/gemm8.o
objs=""
for f in gemm gemm2 kernels attention peav_kernels vit_kernels t5_kernels engine peav vit t5 mbert api; do objs="$objs build/$f.o"; done
hipcc --offload-arch=gfx950 -shared -fPIC $objs ../../tools/abl/gemm8.o -o ../../tools/abl/libsamaudio_hip_abl.so
echo "built tools/abl/libsamaudio_hip_abl.so"
