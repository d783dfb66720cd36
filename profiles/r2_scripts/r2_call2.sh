#!/bin/bash
# Round 2, GPU call 2: new GEMM policy (8-phase for wide outputs, loader-wave persistent 256x128 elsewhere), the parity
# tests at the benchmarked dims, the new bench line (attributable rooflines + parity_check), A/B vs the round-1 policy,
# streams, small per-GPU batches, BASELINE configs[1], a rocprofv3 kernel trace of the same command.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r2_call2
mkdir -p $OUT
(timeout 900 python -m pytest tests -m gpu -q -x -s) > $OUT/gpu_tests.log 2>&1; tail -3 $OUT/gpu_tests.log
(timeout 400 python bench.py) > $OUT/bench.log 2>&1; tail -1 $OUT/bench.log | cut -c1-600
(SAMAUDIO_DEBUG_FLAGS=5=1 timeout 300 python bench.py --no-cpu-baseline --no-roofline) > $OUT/bench_r1policy.log 2>&1; tail -1 $OUT/bench_r1policy.log | cut -c1-200
(timeout 300 python bench.py --no-cpu-baseline --no-roofline --streams 2) > $OUT/bench_streams2.log 2>&1; tail -1 $OUT/bench_streams2.log | cut -c1-200
(timeout 300 python bench.py --no-cpu-baseline --batch 4 --graph 0 --steps 5) > $OUT/bench_b4.log 2>&1; tail -1 $OUT/bench_b4.log | cut -c1-200
(timeout 300 python bench.py --no-cpu-baseline --size 'small*' --batch 8 --graph 0 --steps 5) > $OUT/bench_small_b8.log 2>&1; tail -1 $OUT/bench_small_b8.log | cut -c1-200
(timeout 300 python tools/gemm_bench.py --experimental --batch 4) > $OUT/gemm_m1000.log 2>&1; tail -6 $OUT/gemm_m1000.log | cut -c1-300
(timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline) > $OUT/trace.log 2>&1; echo trace exit=$?
python tools/rocpd_stats.py $(find $OUT/trace -name '*results.db' | head -1) > $OUT/kernel_stats.md 2>$OUT/kernel_stats.err; head -30 $OUT/kernel_stats.md
rm -rf $OUT/trace
ls -la $OUT
