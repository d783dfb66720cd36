#!/bin/bash
# Builds libsamaudio_hip.so for gfx950 (hipcc cross-compiles without a GPU).  In-tree output so the
# .so travels with the repo snapshot to the GPU box.
set -e
cd "$(dirname "$0")"
OUT=../libsamaudio_hip.so
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result"
mkdir -p build
pids=()
for f in gemm gemm2 gemm8 kernels attention peav_kernels vit_kernels engine peav vit api; do
  if [ ! -f build/$f.o ] || [ $f.hip -nt build/$f.o ] || [ common.h -nt build/$f.o ] || [ kernels.h -nt build/$f.o ] || [ engine.h -nt build/$f.o ] || [ peav.h -nt build/$f.o ] || [ vit.h -nt build/$f.o ] || [ ../../include/samaudio.h -nt build/$f.o ]; then
    EXTRA=""
    # gemm2.hip: the fully unrolled 4x4-fragment epilogue exceeds clang's default pragma-unroll budget; without the
    # full unroll the accumulator array is indexed dynamically and lands in scratch memory.
    if [ $f = gemm2 ] || [ $f = gemm8 ]; then EXTRA="-mllvm -pragma-unroll-threshold=262144"; fi
    hipcc $FLAGS $EXTRA -c $f.hip -o build/$f.o &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC build/gemm.o build/gemm2.o build/gemm8.o build/kernels.o build/attention.o build/peav_kernels.o build/vit_kernels.o build/engine.o build/peav.o build/vit.o build/api.o -o $OUT
echo "built $OUT"
