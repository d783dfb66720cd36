#!/bin/bash
# Round 2, GPU call 10: each row group decodes its own waveforms on its stream (codec beside the other group's kernels),
# the kept part of call 9 (gemm8 epilogue with loads ahead of stores; gemm2-family epilogues, attention prefetch and fold
# double-buffering reverted: measured slower / no gain), whole -m gpu suite.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r2_call10
mkdir -p $OUT
PREV=$GRAFT_REPO_ROOT/tools/_ab/libsamaudio_hip_prev.so
(timeout 900 python -m pytest tests -m gpu -q) > $OUT/gpu_tests.log 2>&1; tail -3 $OUT/gpu_tests.log
b() { name=$1; shift; (timeout 400 python bench.py --no-cpu-baseline "$@") > $OUT/bench_$name.log 2>&1; tail -1 $OUT/bench_$name.log | cut -c1-160; }
b new_s2
SAMAUDIO_LIB_AB=$PREV b prevlib_s2 --no-roofline
b new_s2_again --no-roofline
b new_s1 --streams 1 --no-roofline
b new_visual_b4 --visual --batch 4 --steps 3 --no-roofline
b new_rerank_b8 --batch 8 --candidates 8 --predict-spans --steps 2 --no-roofline
b new_fp16 --precision fp16 --no-roofline
ls $OUT
