#!/bin/bash
# Round 2, GPU call 6: the fp16-operand build (libsamaudio_hip_f16.so) - GEMM kernels on fp16 operands, separate() and the
# large*-dims forward / 2-step solve against the oracle, the bench line in fp16 next to bf16; 8-phase family from N >= 1024
# (vision tower out_proj / c_proj, small*).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r2_call6
mkdir -p $OUT
(timeout 900 python -m pytest tests -m gpu -q -x -s) > $OUT/gpu_tests.log 2>&1; tail -3 $OUT/gpu_tests.log; grep -E "fp16" $OUT/gpu_tests.log | cut -c1-220 | head -30
b() { name=$1; shift; (timeout 400 python bench.py --no-cpu-baseline "$@") > $OUT/bench_$name.log 2>&1; tail -1 $OUT/bench_$name.log | cut -c1-200; grep -E "vision tower:|parity_check:" $OUT/bench_$name.log | cut -c1-400; }
b fp16 --precision fp16 --verify
b bf16 --verify
b visual_b4 --visual --batch 4 --steps 3
b visual_b4_fp16 --visual --batch 4 --steps 3 --precision fp16 --no-roofline
b small_b8 --size 'small*' --batch 8 --steps 5
ls -la $OUT
