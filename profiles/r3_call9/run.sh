#!/bin/bash
# GPU call 9 of round 3: the pipelined gemm8s with a 4-stage ring (3 K-tiles of L2 latency in flight instead of 2) for
# launches of <= 256 workgroups - A/B against the 3-stage ring (flag 22) on the same box: GEMM sweep at 4 / 8 clips, the
# strong-scaling share (4 clips), small* 8 clips (configs[1]), configs[3] shape unaffected; hardware tests of the family.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3_call9
mkdir -p $O
( timeout 600 python -m pytest tests/test_gemm2_gpu.py tests/test_fp16_gpu.py -m gpu -x -q ) > $O/gpu_tests_subset.log 2>&1
( timeout 300 python tools/gemm_bench.py --clips 4 8 --iters 20 --no-blas ) > $O/gemm_bench_s4.log 2>&1
( SAMAUDIO_DEBUG_FLAGS=22=1 timeout 300 python tools/gemm_bench.py --clips 4 8 --iters 20 --no-blas ) > $O/gemm_bench_s3.log 2>&1
Q="--no-cpu-baseline --no-parity-mode"
( timeout 300 python bench.py $Q --batch 4 --steps 6 --warmup 2 ) > $O/bench_batch4_s4.log 2>&1
( SAMAUDIO_DEBUG_FLAGS=22=1 timeout 300 python bench.py $Q --batch 4 --steps 6 --warmup 2 ) > $O/bench_batch4_s3.log 2>&1
( timeout 300 python bench.py $Q --size 'small*' --batch 8 --steps 6 --warmup 2 ) > $O/bench_small_b8_s4.log 2>&1
( SAMAUDIO_DEBUG_FLAGS=22=1 timeout 300 python bench.py $Q --size 'small*' --batch 8 --steps 6 --warmup 2 ) > $O/bench_small_b8_s3.log 2>&1
tail -2 $O/gpu_tests_subset.log
for f in bench_batch4_s4 bench_batch4_s3 bench_small_b8_s4 bench_small_b8_s3; do echo $f; grep -o '"value": [0-9.]*' $O/$f.log | head -1; done
for f in gemm_bench_s4 gemm_bench_s3; do echo $f; grep -v amdgpu $O/$f.log | sed 's/ok  *([^)]*)/ok/g' | grep "M=1000\|M=2000" | cut -c1-250; done
