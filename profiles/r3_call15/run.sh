#!/bin/bash
# GPU call 15 of round 3: gemm.hip (fp32 / generic kernel) with the MFMA operands swapped and a 16-byte epilogue.  Its tests,
# the quick bench line (rows dit/gemm_f32_*: the fp32 classes of the 16-bit mode), and the fp32 parity mode's own rate.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3_call15
mkdir -p $O
( timeout 600 python -m pytest tests/test_gemm_gpu.py tests/test_kernels_gpu.py tests/test_path_gpu.py tests/test_precision_gpu.py tests/test_large_gpu.py -m gpu -q ) > $O/gpu_tests_subset.log 2>&1; echo "tests exit=$?"; tail -3 $O/gpu_tests_subset.log
Q="--no-cpu-baseline --no-parity-mode --steps 6 --warmup 2"
( timeout 300 python bench.py $Q ) > $O/bench.log 2>&1
grep -o '"value": [0-9.]*' $O/bench.log | head -1
( timeout 400 python bench.py --no-cpu-baseline --no-parity-mode --precision fp32 --batch 8 --steps 2 --warmup 1 ) > $O/bench_fp32_b8.log 2>&1
grep -o '"value": [0-9.]*' $O/bench_fp32_b8.log | head -1
