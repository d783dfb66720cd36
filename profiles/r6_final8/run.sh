#!/bin/bash
# Last check of the round's final build (the .so files rebuilt after the comment-only header edit): smoke(), the GEMM and x3 test files,
# the default bench line.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${R6_OUT:-r6_final8}; mkdir -p $O
export OMP_NUM_THREADS=16
( time timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ) > $O/smoke.log 2>&1; grep "smoke" $O/smoke.log | cut -c1-120
( timeout 1500 python -m pytest tests/test_gemm2_gpu.py tests/test_gemm_gpu.py tests/test_x3_gpu.py tests/test_path_gpu.py -x -q -m gpu -p no:cacheprovider ) > $O/gpu_tests_subset.log 2>&1; echo "subset exit=$?"; tail -2 $O/gpu_tests_subset.log
( time timeout 900 python bench.py --no-other-configs ) > $O/bench_default.log 2> $O/bench_default.err; echo "bench exit=$?"; grep -o '"value": [0-9.]*' $O/bench_default.log | head -3
