#!/bin/bash
# Round 2, GPU call 26: the whole -m gpu suite on the final code (after the pipelined gemm8s of call 25), smoke(), and the
# bench lines of the two configurations that kernel changes (4 clips, small*) with their rooflines.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r2_call26
mkdir -p $OUT
(timeout 900 python -m pytest tests -m gpu -q -s) > $OUT/gpu_tests.log 2>&1; tail -2 $OUT/gpu_tests.log
(timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')") > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
b() { name=$1; shift; (timeout 400 python bench.py --no-cpu-baseline "$@") > $OUT/bench_$name.log 2>&1; echo "$name $(tail -1 $OUT/bench_$name.log | cut -c52-100)"; }
b batch4 --batch 4 --steps 5
b small_b8 --size 'small*' --batch 8 --steps 5
b rerank_b8 --batch 8 --candidates 8 --predict-spans --steps 2
b visual_b4 --visual --batch 4 --steps 3
