#!/bin/bash
# Round 2, GPU call 16: vision tower on two streams (A/B by SAMAUDIO_VIT_STREAMS), tail split restricted to <= 4 rounds.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r2_call16
mkdir -p $OUT
(timeout 600 python -m pytest tests/test_vit_gpu.py tests/test_gemm2_gpu.py tests/test_path_gpu.py -m gpu -q) > $OUT/gpu_tests_subset.log 2>&1; tail -2 $OUT/gpu_tests_subset.log
b() { name=$1; shift; (timeout 400 python bench.py --no-cpu-baseline "$@") > $OUT/bench_$name.log 2>&1; tail -1 $OUT/bench_$name.log | cut -c52-100; grep "vision tower:" $OUT/bench_$name.log; }
b visual_2s --visual --batch 4 --steps 3 --no-roofline
SAMAUDIO_VIT_STREAMS=1 b visual_1s --visual --batch 4 --steps 3 --no-roofline
b visual_2s_again --visual --batch 4 --steps 3 --no-roofline
b default --no-roofline
(timeout 300 python tools/gemm_bench.py --vit) > $OUT/gemm_vit.log 2>&1; tail -4 $OUT/gemm_vit.log | cut -c1-200
