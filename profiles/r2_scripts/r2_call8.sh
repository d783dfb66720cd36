#!/bin/bash
# Round 2, GPU call 8: loads issued ahead of stores in the GEMM epilogues (gemm8 / gemm8s epilogue8, the gemm2-family
# epilogues) and in the cross-attention fold kernel (vmcnt retires in issue order: a load behind a store waits for the
# store's round trip); op-level timings, GEMM family table, bench lines (1 and 2 streams), T5 inside the step.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r2_call8
mkdir -p $OUT
(timeout 600 python -m pytest tests/test_gemm2_gpu.py tests/test_gemm_gpu.py tests/test_kernels_gpu.py tests/test_path_gpu.py tests/test_large_gpu.py tests/test_zz_next_rows_gpu.py -m gpu -q) > $OUT/gpu_tests_subset.log 2>&1; tail -2 $OUT/gpu_tests_subset.log
(timeout 300 python tools/op_bench.py) > $OUT/op_bench.log 2>&1; tail -16 $OUT/op_bench.log
(timeout 200 python tools/gemm_bench.py --family) > $OUT/gemm_family_m8000.log 2>&1; tail -6 $OUT/gemm_family_m8000.log | cut -c1-200
b() { name=$1; shift; (timeout 400 python bench.py --no-cpu-baseline "$@") > $OUT/bench_$name.log 2>&1; tail -1 $OUT/bench_$name.log | cut -c1-160; }
b s2
b s1 --streams 1
b s2_t5 --t5 --no-roofline
b visual_b4 --visual --batch 4 --steps 3 --no-roofline
b rerank_b8 --batch 8 --candidates 8 --predict-spans --steps 2
ls $OUT
