#!/bin/bash
# Round 2, GPU call 13: which tile family suits the 96-channel decoder stage (T = 480 000) - microbenchmark only.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r2_call13
mkdir -p $OUT
(timeout 400 python tools/op_bench.py) > $OUT/op_bench.log 2>&1; grep -E "codec" $OUT/op_bench.log
