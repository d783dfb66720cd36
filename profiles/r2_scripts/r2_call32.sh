#!/bin/bash
# Round 2, GPU call 32: probe - what do the modulation table loads of rmsnorm_mod cost (the kernel with and without them)?
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r2_call32
mkdir -p $OUT
(timeout 300 python tools/op_bench.py) > $OUT/op_bench.log 2>&1; grep -E "rmsnorm|qkv_prep" $OUT/op_bench.log
