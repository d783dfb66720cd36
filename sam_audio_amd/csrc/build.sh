#!/bin/bash
# Builds libsamaudio_hip.so (16-bit operands = bf16) and libsamaudio_hip_f16.so (the same sources with -DSA_OPERAND_FP16:
# 16-bit operands = IEEE fp16, the precision="fp16" mode) for gfx950; hipcc cross-compiles without a GPU.  In-tree output
# so the .so files travel with the repo snapshot to the GPU box.
set -e
cd "$(dirname "$0")"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result"
# experiment switch (common.h f2bf): the bf16 build rounds with the hardware conversion; separate object directory
HW=""; HWDIR=""
if [ -n "$SAMAUDIO_BF16_HW_ROUND" ]; then HW="-DSA_BF16_HW_ROUND"; HWDIR="_hwround"; fi
SRCS="gemm gemm2 gemm8 kernels attention peav_kernels vit_kernels t5_kernels engine peav vit t5 mbert api"
build_one() {  # $1 = object dir, $2 = extra flags, $3 = output
  local dir=$1 extra_all=$2 out=$3 pids=() objs=""
  mkdir -p $dir
  for f in $SRCS; do
    objs="$objs $dir/$f.o"
    if [ ! -f $dir/$f.o ] || [ $f.hip -nt $dir/$f.o ] || [ common.h -nt $dir/$f.o ] || [ kernels.h -nt $dir/$f.o ] || [ engine.h -nt $dir/$f.o ] || [ peav.h -nt $dir/$f.o ] || [ vit.h -nt $dir/$f.o ] || [ t5.h -nt $dir/$f.o ] || [ mbert.h -nt $dir/$f.o ] || [ ../../include/samaudio.h -nt $dir/$f.o ] || [ build.sh -nt $dir/$f.o ]; then
      EXTRA=""
      # gemm2.hip / gemm8.hip: the fully unrolled 4x4-fragment epilogues exceed clang's default pragma-unroll budget; without
      # the full unroll the accumulator array is indexed dynamically and lands in scratch memory.
      if [ $f = gemm2 ] || [ $f = gemm8 ]; then EXTRA="-mllvm -pragma-unroll-threshold=262144 -Wno-inline-asm"; fi
      if [ $f = gemm ]; then EXTRA="-mllvm -pragma-unroll-threshold=262144 -Wno-inline-asm"; fi   # (gemm.hip: the 128 x 128 instantiations kept their accumulators in scratch memory without it - round 6, profiles/r6_call13/)
      hipcc $FLAGS $extra_all $EXTRA -c $f.hip -o $dir/$f.o &
      pids+=($!)
    fi
  done
  for p in "${pids[@]}"; do wait $p; done
  hipcc --offload-arch=gfx950 -shared -fPIC $objs -o $out
  echo "built $out"
}
build_one build$HWDIR "$HW" ../libsamaudio_hip$HWDIR.so &   # (the experiment build: libsamaudio_hip_hwround.so, loaded with SAMAUDIO_LIB_AB)
B1=$!
build_one build_f16 "-DSA_OPERAND_FP16" ../libsamaudio_hip_f16.so &
B2=$!
wait $B1
wait $B2
