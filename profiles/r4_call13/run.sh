#!/bin/bash
# Round-4 call 13: (a) the pipelined gemm8s form with requesting waves (debug flag 27 = 2 / 3) against the form without (1):
# bitwise tests, per-launch times on the few-row shapes (4 / 2 / 1 clips per launch), end to end at 4 clips and at small* 8 clips;
# (b) the two-stream scenario 300 times in both allocator modes; (c) the new fp32-copy tests.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4_call13; mkdir -p $O
export OMP_NUM_THREADS=16
( timeout 600 python -m pytest tests/test_gemm2_gpu.py tests/test_fp16_gpu.py tests/test_precision_gpu.py -m gpu -q -p no:cacheprovider -k "pipelined_form or wave_roles or mixed_mode_gemm or option_validation or f32_copies" ) > $O/tests_roles.log 2>&1; echo "tests exit=$?"; tail -2 $O/tests_roles.log
timeout 400 python tools/gemm_bench.py --roles --clips 4 2 1 --iters 20 > $O/gemm_bench_roles.log 2>&1; grep -c "WRONG" $O/gemm_bench_roles.log
Q="--no-cpu-baseline --no-parity-mode --no-other-configs --no-roofline --steps 6 --warmup 2"
for r in 1 2 3; do
  SAMAUDIO_DEBUG_FLAGS=27=$r timeout 300 python bench.py $Q --batch 4 > $O/bench_b4_roles$r.log 2>&1; echo "4 clips, flag 27=$r: $(grep -o '"value": [0-9.]*' $O/bench_b4_roles$r.log | tail -1)"
done
for r in 1 2; do
  SAMAUDIO_DEBUG_FLAGS=27=$r timeout 300 python bench.py $Q --size 'small*' --batch 8 > $O/bench_small_roles$r.log 2>&1; echo "small* 8 clips, flag 27=$r: $(grep -o '"value": [0-9.]*' $O/bench_small_roles$r.log | tail -1)"
done
SAMAUDIO_POISON=1 timeout 400 python tools/stress_two_streams.py --reps 300 > $O/stress_two_streams_caching.log 2>&1; tail -1 $O/stress_two_streams_caching.log
PYTORCH_NO_HIP_MEMORY_CACHING=1 PYTORCH_NO_CUDA_MEMORY_CACHING=1 timeout 500 python tools/stress_two_streams.py --reps 300 > $O/stress_two_streams_nocache.log 2>&1; tail -1 $O/stress_two_streams_nocache.log
( timeout 300 python -m pytest tests/test_path_gpu.py -m gpu -q -p no:cacheprovider ) > $O/tests_path.log 2>&1; tail -1 $O/tests_path.log
