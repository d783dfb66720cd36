#!/bin/bash
# Round 2, GPU call 27 (NEGATIVE RESULT: 57.2 vs 49.2 us; the candidate kernel was removed again - check out commit
# "cross_attn_fold3 (A/B candidate)" to reproduce): cross_attn_fold with the V rows of a trip staged once per workgroup in LDS (flag 0 = 2) - test on
# hardware, micro-benchmark, bench A/B.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r2_call27
mkdir -p $OUT
(timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q) > $OUT/gpu_tests_subset.log 2>&1; tail -2 $OUT/gpu_tests_subset.log
(timeout 300 python tools/op_bench.py) > $OUT/op_bench.log 2>&1; grep -E "cross_attn_fold" $OUT/op_bench.log
b() { name=$1; shift; (timeout 400 python bench.py --no-cpu-baseline --no-roofline "$@") > $OUT/bench_$name.log 2>&1; echo "$name $(tail -1 $OUT/bench_$name.log | cut -c52-100)"; }
b default
SAMAUDIO_DEBUG_FLAGS=0=2 b fold_lds_v
b default_again
SAMAUDIO_DEBUG_FLAGS=0=2 b fold_lds_v_again
