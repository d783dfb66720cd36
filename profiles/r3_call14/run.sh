#!/bin/bash
# GPU call 14 of round 3: Snake of the 16-bit epilogues on the hardware reciprocal (snake16_f) instead of an IEEE division.
# Residual units in isolation, the tests that pin the codec / the full solve to the oracle, the quick bench line.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3_call14
mkdir -p $O
( timeout 300 python tools/op_bench.py ) > $O/op_bench.log 2>&1
grep "residual unit" $O/op_bench.log
( timeout 900 python -m pytest tests/test_gemm2_gpu.py tests/test_path_gpu.py tests/test_large_gpu.py tests/test_fp16_gpu.py tests/test_precision_gpu.py -m gpu -q ) > $O/gpu_tests_subset.log 2>&1; echo "tests exit=$?"; tail -3 $O/gpu_tests_subset.log
Q="--no-cpu-baseline --no-parity-mode --steps 6 --warmup 2"
( timeout 300 python bench.py $Q ) > $O/bench.log 2>&1
grep -o '"value": [0-9.]*' $O/bench.log | head -1
grep -o '"parity_check": {[^}]*}' $O/bench.log | head -1
