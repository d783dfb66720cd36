#!/bin/bash
# Round 2, GPU call 28: self-attention with the query blocks of a (batch, head) dealt onto one XCD (flag 23): bitwise test on
# hardware, micro-benchmark, bench A/B on configs[2] and on the vision tower (5 query blocks per (frame, head)).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r2_call28
mkdir -p $OUT
(timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q) > $OUT/gpu_tests_subset.log 2>&1; tail -2 $OUT/gpu_tests_subset.log
(timeout 300 python tools/op_bench.py) > $OUT/op_bench.log 2>&1; grep -E "self_attention" $OUT/op_bench.log
b() { name=$1; shift; (timeout 400 python bench.py --no-cpu-baseline --no-roofline "$@") > $OUT/bench_$name.log 2>&1; echo "$name $(tail -1 $OUT/bench_$name.log | cut -c52-100) $(grep -o 'vision tower: [0-9.]* ms' $OUT/bench_$name.log | tail -1)"; }
b default
SAMAUDIO_DEBUG_FLAGS=23=1 b attn_xcd
b default_again
SAMAUDIO_DEBUG_FLAGS=23=1 b attn_xcd_again
b visual --visual --batch 4 --steps 3
SAMAUDIO_DEBUG_FLAGS=23=1 b visual_attn_xcd --visual --batch 4 --steps 3
