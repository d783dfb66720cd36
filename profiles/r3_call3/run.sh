#!/bin/bash
# GPU call 3 of round 3: the 8-phase K loop with one staging instruction of every phase moved into the wave's own MFMA
# cluster + the plain-GEMM specialisation (no tap walk) - A/B against the round-2 loop in the same library; the rewritten
# cross_attn_fold kernel (64 channels x 8 heads per workgroup, whole-line stores); tests; quick bench lines.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3_call3
mkdir -p $O
export OMP_NUM_THREADS=16
( timeout 600 python tools/gemm_bench.py --r3 --iters 20 ) > $O/gemm_bench_r3.log 2>&1
( timeout 300 python tools/fold_bench.py ) > $O/fold_bench.log 2>&1
( time timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_gemm2_gpu.py tests/test_path_gpu.py tests/test_fp16_gpu.py -m gpu -x -q ) > $O/gpu_tests_subset.log 2>&1
Q="--no-cpu-baseline --no-parity-mode --steps 4 --warmup 1"
( timeout 300 python bench.py $Q ) > $O/bench_quick.log 2>&1
( timeout 300 python bench.py $Q --streams 1 ) > $O/bench_streams1.log 2>&1
( timeout 300 python bench.py $Q --precision fp16 ) > $O/bench_fp16.log 2>&1
tail -3 $O/gpu_tests_subset.log; cat $O/fold_bench.log | grep fold
for f in bench_quick bench_streams1 bench_fp16; do echo $f; grep -o '"value": [0-9.]*' $O/$f.log | head -1; done
