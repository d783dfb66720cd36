#!/bin/bash
# Round-4 call 14: (a) gemm8s' pipelined form with requesting waves and the L2 prefetch (debug flag 27 = 4 / 5) - bitwise tests, launch
# times, end to end at 4 clips / small* 8 clips; (b) the two-stream difference of call 13 (first call of a freshly built model, allocator
# without caching): is it garbage in a fresh workspace?  finite large-magnitude fill bytes, a fresh model every repetition, and the
# checksum trace naming the first stage that differs.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4_call14; mkdir -p $O
export OMP_NUM_THREADS=16
( timeout 600 python -m pytest tests/test_gemm2_gpu.py tests/test_fp16_gpu.py -m gpu -q -p no:cacheprovider -k "pipelined_form or wave_roles or mixed_mode_gemm" ) > $O/tests_roles.log 2>&1; echo "tests exit=$?"; tail -1 $O/tests_roles.log
timeout 300 python tools/gemm_bench.py --roles --clips 4 --iters 20 > $O/gemm_bench_roles.log 2>&1; echo "WRONG: $(grep -c WRONG $O/gemm_bench_roles.log)"
Q="--no-cpu-baseline --no-parity-mode --no-other-configs --no-roofline --steps 6 --warmup 2"
for r in 1 2 4 5; do
  SAMAUDIO_DEBUG_FLAGS=27=$r timeout 300 python bench.py $Q --batch 4 > $O/bench_b4_roles$r.log 2>&1; echo "4 clips, flag 27=$r: $(grep -o '"value": [0-9.]*' $O/bench_b4_roles$r.log | tail -1)"
done
for r in 1 4 5; do
  SAMAUDIO_DEBUG_FLAGS=27=$r timeout 300 python bench.py $Q --size 'small*' --batch 8 > $O/bench_small_roles$r.log 2>&1; echo "small* 8 clips, flag 27=$r: $(grep -o '"value": [0-9.]*' $O/bench_small_roles$r.log | tail -1)"
done
for b in 0x46 0x7F 0xC6; do
  SAMAUDIO_POISON=1 SAMAUDIO_POISON_BYTE=$b timeout 300 python tools/stress_two_streams.py --reps 60 --rebuild 1 > $O/stress_fill_$b.log 2>&1; echo "fill $b: $(tail -1 $O/stress_fill_$b.log)"
done
PYTORCH_NO_HIP_MEMORY_CACHING=1 PYTORCH_NO_CUDA_MEMORY_CACHING=1 timeout 400 python tools/stress_two_streams.py --reps 200 --rebuild 2 > $O/stress_nocache_rebuild2.log 2>&1; tail -1 $O/stress_nocache_rebuild2.log
SAMAUDIO_POISON=1 SAMAUDIO_POISON_BYTE=0x46 timeout 300 python tools/diag_hash.py --reps 12 --rebuild 1 > $O/diag_hash_fill46.log 2>&1; tail -4 $O/diag_hash_fill46.log
PYTORCH_NO_HIP_MEMORY_CACHING=1 PYTORCH_NO_CUDA_MEMORY_CACHING=1 timeout 400 python tools/diag_hash.py --reps 120 --rebuild 1 > $O/diag_hash_nocache.log 2>&1; tail -4 $O/diag_hash_nocache.log
