#!/bin/bash
# GPU call 8 of round 3: gemm8s on 96 x 128 tiles for launches with few rows (4 clips per GPU: 242 instead of 176
# workgroups at N = D) - hardware tests, the GEMM sweep at 4 / 8 clips, the strong-scaling share and small* (configs[1]),
# each against flag 21 (= the 128-row forms only) on the same box; default line for regression.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3_call8
mkdir -p $O
( timeout 600 python -m pytest tests/test_gemm2_gpu.py tests/test_fp16_gpu.py tests/test_path_gpu.py -m gpu -x -q ) > $O/gpu_tests_subset.log 2>&1
( timeout 300 python tools/gemm_bench.py --clips 4 8 --iters 20 ) > $O/gemm_bench_fewrows.log 2>&1
Q="--no-cpu-baseline --no-parity-mode"
( timeout 300 python bench.py $Q --batch 4 --steps 6 --warmup 2 ) > $O/bench_batch4.log 2>&1
( SAMAUDIO_DEBUG_FLAGS=21=1 timeout 300 python bench.py $Q --batch 4 --steps 6 --warmup 2 ) > $O/bench_batch4_flag21.log 2>&1
( timeout 300 python bench.py $Q --size 'small*' --batch 8 --steps 6 --warmup 2 ) > $O/bench_small_b8.log 2>&1
( SAMAUDIO_DEBUG_FLAGS=21=1 timeout 300 python bench.py $Q --size 'small*' --batch 8 --steps 6 --warmup 2 ) > $O/bench_small_b8_flag21.log 2>&1
( timeout 300 python bench.py $Q --steps 4 --warmup 1 ) > $O/bench_quick.log 2>&1
tail -2 $O/gpu_tests_subset.log
for f in bench_batch4 bench_batch4_flag21 bench_small_b8 bench_small_b8_flag21 bench_quick; do echo $f; grep -o '"value": [0-9.]*' $O/$f.log | head -1; done
grep -v amdgpu $O/gemm_bench_fewrows.log | sed 's/ok  *([^)]*)/ok/g' | cut -c1-300 | tail -13
