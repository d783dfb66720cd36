#!/bin/bash
# Round 2, GPU call 9: A/B on ONE box of the library before (tools/_ab/libsamaudio_hip_prev.so = commit 19fae59) and after
# the "loads ahead of stores" epilogues (depth-4 modulo schedule, loop not fully unrolled), the K/V prefetch of the flash
# attention kernel and the register-resident LayerNorm.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r2_call9
mkdir -p $OUT
PREV=$GRAFT_REPO_ROOT/tools/_ab/libsamaudio_hip_prev.so
(timeout 200 python tools/gemm_bench.py --family) > $OUT/gemm_family_new.log 2>&1; tail -6 $OUT/gemm_family_new.log | cut -c1-200
(SAMAUDIO_LIB_AB=$PREV timeout 200 python tools/gemm_bench.py --family) > $OUT/gemm_family_prev.log 2>&1; tail -6 $OUT/gemm_family_prev.log | cut -c1-200
(timeout 300 python tools/op_bench.py) > $OUT/op_bench_new.log 2>&1; tail -14 $OUT/op_bench_new.log
(SAMAUDIO_LIB_AB=$PREV timeout 300 python tools/op_bench.py) > $OUT/op_bench_prev.log 2>&1; tail -14 $OUT/op_bench_prev.log
b() { name=$1; shift; (timeout 400 python bench.py --no-cpu-baseline "$@") > $OUT/bench_$name.log 2>&1; tail -1 $OUT/bench_$name.log | cut -c1-160; }
b new_s2
SAMAUDIO_LIB_AB=$PREV b prev_s2 --no-roofline
b new_s1 --streams 1
SAMAUDIO_LIB_AB=$PREV b prev_s1 --streams 1 --no-roofline
b new_visual_b4 --visual --batch 4 --steps 3 --no-roofline
ls $OUT
