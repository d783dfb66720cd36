#!/bin/bash
# Builds oracle/_simt/libsamaudio_simt.so: EVERY product source (kernels and host orchestration), compiled unchanged
# as plain C++ against the stand-in <hip/hip_runtime.h> of oracle/simt/stub, so that each hipLaunchKernelGGL runs the
# real kernel body on the functional SIMT simulator (simt.cpp).  TEST INFRASTRUCTURE ONLY.
# `build.sh poison` builds the LDS-poisoning variant (libsamaudio_simt_poison.so, see stub/hip/hip_runtime.h).
set -e
cd "$(dirname "$0")"
SRC=../../sam_audio_amd/csrc
OUT=../_simt
LIB=libsamaudio_simt.so
POISON=""
if [ "$1" = "poison" ]; then OUT=../_simt/poison; LIB=libsamaudio_simt_poison.so; POISON="-DSIMT_POISON"; fi
# `build.sh asan`: the same sources under AddressSanitizer (libsamaudio_simt_asan.so).  "Device" memory is host heap memory on
# the simulator, so a kernel that reads or writes past the end of a tensor it was handed - which on the GPU silently picks up
# whatever the allocator left behind the buffer - is reported with the kernel's source line.  Run with
#   LD_PRELOAD=/opt/rocm/lib/llvm/lib/clang/22/lib/linux/libclang_rt.asan-x86_64.so ASAN_OPTIONS=detect_leaks=0:halt_on_error=0 \
#   SAMAUDIO_EMU_DRYRUN=simt SAMAUDIO_SIMT_ASAN=1 python -m pytest tests/test_path_gpu.py -m gpu -k concurrent_streams
SAN=""
if [ "$1" = "asan" ]; then OUT=../_simt/asan; LIB=libsamaudio_simt_asan.so; SAN="-fsanitize=address -fsanitize-recover=address -fno-omit-frame-pointer -g1 -shared-libsan"; fi
mkdir -p $OUT
CXX=/opt/rocm/lib/llvm/bin/clang++
if [ -n "$SAMAUDIO_BF16_HW_ROUND" ]; then POISON="$POISON -DSA_BF16_HW_ROUND"; OUT=${OUT}_hwround; LIB=${LIB%.so}_hwround.so; mkdir -p $OUT; fi   # experiment build (common.h f2bf); run with SAMAUDIO_SIMT_LIB=oracle/_simt/<that library>
FLAGS="-x c++ -std=c++17 -O2 -fPIC -fopenmp $POISON $SAN -I stub -Wno-unused-value -Wno-unknown-attributes -Wno-ignored-attributes -Wno-pass-failed -Wno-keyword-macro -Wno-psabi"
pids=()
for f in gemm gemm2 gemm8 kernels attention peav_kernels vit_kernels t5_kernels engine peav vit t5 mbert api; do
  src=$SRC/$f.hip
  if [ ! -f $OUT/$f.o ] || [ $src -nt $OUT/$f.o ] || [ stub/hip/hip_runtime.h -nt $OUT/$f.o ] || [ $SRC/common.h -nt $OUT/$f.o ] || [ $SRC/kernels.h -nt $OUT/$f.o ] || [ $SRC/engine.h -nt $OUT/$f.o ] || [ $SRC/peav.h -nt $OUT/$f.o ] || [ $SRC/vit.h -nt $OUT/$f.o ] || [ $SRC/t5.h -nt $OUT/$f.o ] || [ $SRC/mbert.h -nt $OUT/$f.o ]; then
    EXTRA=""
    if [ $f = gemm2 ]; then
      EXTRA="-O1 -I $SRC"   # the fully unrolled epilogues are slow to optimise on the host
      # the one inline-asm statement whose vmcnt count is an asm OPERAND (invisible to the preprocessor) is rewritten to
      # the simulator's call; everything else in the file is compiled as it stands
      sed 's|asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");|simt::wait_vmcnt(N);|' $src > $OUT/gemm2_simt.hip
      grep -q "simt::wait_vmcnt(N);" $OUT/gemm2_simt.hip || { echo "gemm2.hip: wait_vmcnt<N> statement not found"; exit 1; }
      # and the direct-to-LDS load, inline assembly with operands in the product (see dma16a), becomes the builtin the stub emulates
      sed -i 's|^.*// SIMT-DMA$|  simt::dma_asm = true; __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc, (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0); simt::dma_asm = false; (void)lds;|' $OUT/gemm2_simt.hip
      grep -q "(void)lds;" $OUT/gemm2_simt.hip || { echo "gemm2.hip: dma16a asm statement not found"; exit 1; }
      src=$OUT/gemm2_simt.hip
    fi
    if [ $f = gemm8 ]; then
      # gemm8.hip's scalar-base DMA (dma16s: inline assembly with operands) becomes the builtin the stub emulates
      sed 's|^.*// SIMT-DMA8$|  simt::dma_asm = true; __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)((const char*)sbase + voff), (__attribute__((address_space(3))) void*)lds_wave_addr, 16, 0, 0); simt::dma_asm = false; (void)lds;|' $src > $OUT/gemm8_simt.hip
      grep -q "(void)lds;" $OUT/gemm8_simt.hip || { echo "gemm8.hip: dma16s asm statement not found"; exit 1; }
      sed -i 's|^.*// SIMT-DMA8V$|  simt::dma_asm = true; __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc, (__attribute__((address_space(3))) void*)lds_wave_addr, 16, 0, 0); simt::dma_asm = false; (void)lds; /*v*/|' $OUT/gemm8_simt.hip
      grep -q "(void)lds; /\*v\*/" $OUT/gemm8_simt.hip || { echo "gemm8.hip: dma16v asm statement not found"; exit 1; }
      EXTRA="-I $SRC"
      src=$OUT/gemm8_simt.hip
    fi
    $CXX $FLAGS $EXTRA -c $src -o $OUT/$f.o &
    pids+=($!)
  fi
done
$CXX $FLAGS -c simt.cpp -o $OUT/simt.o &
pids+=($!)
for p in "${pids[@]}"; do wait $p; done
$CXX -shared -fPIC -fopenmp $SAN $OUT/gemm.o $OUT/gemm2.o $OUT/gemm8.o $OUT/kernels.o $OUT/attention.o $OUT/peav_kernels.o $OUT/vit_kernels.o $OUT/t5_kernels.o $OUT/engine.o $OUT/peav.o $OUT/vit.o $OUT/t5.o $OUT/mbert.o $OUT/api.o $OUT/simt.o -o ../_simt/$LIB.tmp
mv -f ../_simt/$LIB.tmp ../_simt/$LIB   # atomic: a process that has the old library mapped keeps its inode
echo "built ../_simt/$LIB"
