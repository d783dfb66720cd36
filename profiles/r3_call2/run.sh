#!/bin/bash
# GPU call 2 of round 3: the 256x192 tile of the 8-phase kernel (gemm8n) - forced-variant sweep beside hipBLASLt, the bench
# line with the new tile policy vs the old one (flag 26) vs one row group, the GEMM / path / precision tests on hardware,
# and the full default bench line (parity mode side by side).
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3_call2
mkdir -p $O
export OMP_NUM_THREADS=16
( time timeout 600 python tools/gemm_bench.py --r3 --iters 10 ) > $O/gemm_bench_r3.log 2>&1
( time timeout 900 python -m pytest tests/test_gemm2_gpu.py tests/test_gemm_gpu.py tests/test_path_gpu.py tests/test_precision_gpu.py -m gpu -x -q ) > $O/gpu_tests_subset.log 2>&1
Q="--no-cpu-baseline --no-parity-mode --steps 4 --warmup 1"
( timeout 300 python bench.py $Q ) > $O/bench_new_policy.log 2>&1
( SAMAUDIO_DEBUG_FLAGS=26=1 timeout 300 python bench.py $Q ) > $O/bench_flag26_old_policy.log 2>&1
( timeout 300 python bench.py $Q --streams 1 ) > $O/bench_streams1.log 2>&1
( SAMAUDIO_DEBUG_FLAGS=26=1 timeout 300 python bench.py $Q --streams 1 ) > $O/bench_streams1_flag26.log 2>&1
( timeout 300 python bench.py $Q --batch 4 ) > $O/bench_batch4.log 2>&1
( time timeout 900 python bench.py ) > $O/bench.log 2>&1
tail -3 $O/gpu_tests_subset.log
for f in bench_new_policy bench_flag26_old_policy bench_streams1 bench_streams1_flag26 bench_batch4 bench; do echo $f; grep -o '"value": [0-9.]*' $O/$f.log | head -1; done
