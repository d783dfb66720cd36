#!/bin/bash
# Round 2, GPU call 29: the whole -m gpu suite, smoke() and the default bench line (with CPU baseline and parity check) on
# the final code of the round.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r2_call29
mkdir -p $OUT
(timeout 900 python -m pytest tests -m gpu -q -s) > $OUT/gpu_tests.log 2>&1; tail -2 $OUT/gpu_tests.log
(timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')") > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
(timeout 500 python bench.py) > $OUT/bench.log 2>&1; tail -1 $OUT/bench.log | cut -c1-200
