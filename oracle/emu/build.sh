#!/bin/bash
# Builds oracle/_emu/libsamaudio_emu.so: the product's HOST orchestration sources (engine.hip, peav.hip, api.hip),
# compiled unchanged in host-only mode, linked against the CPU emulation of the kernel launchers (emu_kernels.cpp) and
# host stand-ins for the HIP runtime calls they make (emu_hip.cpp).  TEST INFRASTRUCTURE ONLY - see emu_kernels.cpp.
set -e
cd "$(dirname "$0")"
SRC=../../sam_audio_amd/csrc
OUT=../_emu
mkdir -p $OUT
REN=""
for f in hipMemsetAsync hipMemcpyAsync hipMemcpy hipStreamSynchronize hipEventCreate hipEventDestroy hipEventRecord \
         hipEventSynchronize hipEventElapsedTime hipStreamBeginCapture hipStreamCreateWithFlags; do
  REN="$REN -D$f=emu_$f"
done
FLAGS="--offload-host-only --offload-arch=gfx950 -O2 -std=c++17 -fPIC -fopenmp -Wno-unused-result $REN"
pids=()
for f in $SRC/engine.hip $SRC/peav.hip $SRC/api.hip; do
  hipcc $FLAGS -c $f -o $OUT/$(basename $f .hip).o &
  pids+=($!)
done
hipcc $FLAGS -x hip -c emu_kernels.cpp -o $OUT/emu_kernels.o &
pids+=($!)
hipcc $FLAGS -x hip -c emu_hip.cpp -o $OUT/emu_hip.o &
pids+=($!)
for p in "${pids[@]}"; do wait $p; done
hipcc -shared -fPIC -fopenmp $OUT/engine.o $OUT/peav.o $OUT/api.o $OUT/emu_kernels.o $OUT/emu_hip.o -o $OUT/libsamaudio_emu.so.tmp
mv -f $OUT/libsamaudio_emu.so.tmp $OUT/libsamaudio_emu.so   # atomic: a process that has the old library mapped keeps its inode
echo "built $OUT/libsamaudio_emu.so"
