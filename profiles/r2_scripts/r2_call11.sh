#!/bin/bash
# Round 2, GPU call 11: flash attention with DMA double-buffered K / V^T tiles and the key mask staged through LDS (one
# barrier per key tile) - A/B against the previous library on one box; op-level, bench, vision tower; whole -m gpu suite.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r2_call11
mkdir -p $OUT
PREV=$GRAFT_REPO_ROOT/tools/_ab/libsamaudio_hip_prev.so
(timeout 900 python -m pytest tests -m gpu -q) > $OUT/gpu_tests.log 2>&1; tail -3 $OUT/gpu_tests.log
(timeout 300 python tools/op_bench.py) > $OUT/op_bench_new.log 2>&1; grep -E "self_attention|qkv_prep \(16" $OUT/op_bench_new.log
(SAMAUDIO_LIB_AB=$PREV timeout 300 python tools/op_bench.py) > $OUT/op_bench_prev.log 2>&1; grep -E "self_attention|qkv_prep \(16" $OUT/op_bench_prev.log
b() { name=$1; shift; (timeout 400 python bench.py --no-cpu-baseline "$@") > $OUT/bench_$name.log 2>&1; tail -1 $OUT/bench_$name.log | cut -c1-160; grep "vision tower:" $OUT/bench_$name.log; }
b new_s2
SAMAUDIO_LIB_AB=$PREV b prevlib_s2 --no-roofline
b new_s1 --streams 1 --no-roofline
b new_visual_b4 --visual --batch 4 --steps 3 --no-roofline
ls $OUT
