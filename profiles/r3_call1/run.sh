#!/bin/bash
# GPU call 1 of round 3: precision classes (fp32 small GEMMs inside the 16-bit engines), the error budget of the 16-bit
# modes on the full solve at large*, the full-solve parity test, same-box yardsticks, and the bench line on the new default.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3_call1
mkdir -p $O
export OMP_NUM_THREADS=16
( time timeout 900 python -m pytest tests/test_precision_gpu.py tests/test_large_gpu.py -m gpu -x -q -s ) > $O/gpu_tests_precision.log 2>&1
( time timeout 600 python tools/error_budget.py --out $O/error_budget.json ) > $O/error_budget.log 2>&1
( time timeout 300 python tools/yardstick.py gemm --out $O/yardstick.json ) > $O/yardstick_gemm.log 2>&1
( time timeout 420 python tools/yardstick.py eager --out $O/yardstick.json ) > $O/yardstick_eager.log 2>&1
( time timeout 600 python bench.py ) > $O/bench.log 2>&1
( time timeout 400 python bench.py --precision fp16 --no-cpu-baseline ) > $O/bench_fp16.log 2>&1
tail -3 $O/gpu_tests_precision.log; tail -4 $O/error_budget.log; tail -2 $O/bench.log | cut -c1-600
