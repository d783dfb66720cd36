#!/bin/bash
# Round-4 call 17: gemm8s' pipelined form with requesting waves that stage through registers (debug flag 27 = 6: 7 K-tiles in flight
# instead of 3) against the shipped form (1) and the direct-to-LDS roles form (2): bitwise tests, launch times, kernel durations inside
# the model (rocprofv3) and end to end at 4 clips, small* 8 clips and the visual-prompt configuration.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4_call17; mkdir -p $O
export OMP_NUM_THREADS=16
( timeout 600 python -m pytest tests/test_gemm2_gpu.py tests/test_fp16_gpu.py -m gpu -q -p no:cacheprovider -k "pipelined_form or wave_roles or mixed_mode_gemm" ) > $O/tests_roles.log 2>&1; echo "tests exit=$?"; tail -1 $O/tests_roles.log
timeout 300 python tools/gemm_bench.py --roles --clips 4 --iters 20 > $O/gemm_bench_roles.log 2>&1; echo "WRONG: $(grep -c WRONG $O/gemm_bench_roles.log)"
Q="--no-cpu-baseline --no-parity-mode --no-other-configs --no-roofline --steps 6 --warmup 2"
for r in 1 6 1 6; do
  SAMAUDIO_DEBUG_FLAGS=27=$r timeout 300 python bench.py $Q --batch 4 > $O/bench_b4_roles$r.log 2>&1; echo "4 clips, flag 27=$r: $(grep -o '"value": [0-9.]*' $O/bench_b4_roles$r.log | tail -1)"
done
for r in 1 6; do
  SAMAUDIO_DEBUG_FLAGS=27=$r timeout 300 python bench.py $Q --size 'small*' --batch 8 > $O/bench_small_roles$r.log 2>&1; echo "small* 8 clips, flag 27=$r: $(grep -o '"value": [0-9.]*' $O/bench_small_roles$r.log | tail -1)"
done
for r in 1 6; do
  SAMAUDIO_DEBUG_FLAGS=27=$r timeout 300 python bench.py $Q --visual --batch 4 --steps 3 --warmup 1 > $O/bench_visual_roles$r.log 2>&1; echo "visual 4 clips, flag 27=$r: $(grep -o '"value": [0-9.]*' $O/bench_visual_roles$r.log | tail -1)"
done
Q2="--no-cpu-baseline --no-parity-mode --no-other-configs --no-roofline --steps 2 --warmup 1"
( SAMAUDIO_DEBUG_FLAGS=27=6 timeout 300 rocprofv3 --kernel-trace --stats -d $O/t6 -o t -- python bench.py $Q2 --batch 4 ) > $O/trace_b4_roles6.log 2>&1
db=$(find $O/t6 -name '*_results.db' | head -1); [ -n "$db" ] && python tools/rocpd_stats.py $db > $O/kernel_stats_b4_roles6.md 2>/dev/null; rm -rf $O/t6
grep "gemm8" $O/kernel_stats_b4_roles6.md | head -4 | cut -c1-130
