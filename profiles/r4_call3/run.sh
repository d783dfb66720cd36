#!/bin/bash
# round 4, GPU call 3: (a) the two lean epilogues of the 8-phase family - epilogue8_linear (registers) for 16-bit-only outputs,
# epilogue8_rows (through the wave's LDS area) for fp32 output / residual; debug flag 24: 0 = shipped choice, 1 = general,
# 2 / 3 = one form for everything - tests, time vs K, DiT shapes; (b) where a two-stream solve first differs from the
# one-stream solve (tools/diag_hash.py: per-stage checksums on the launch streams, 60 repetitions); (c) end-to-end tests
# and the quick bench lines.
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r4_call3; mkdir -p $O
timeout 900 python -m pytest tests/test_gemm2_gpu.py tests/test_gemm_gpu.py -m gpu -q -p no:cacheprovider > $O/tests_gemm.log 2>&1; tail -1 $O/tests_gemm.log
timeout 900 python tools/diag_hash.py --reps 60 > $O/diag_hash.log 2>&1; tail -8 $O/diag_hash.log
timeout 900 python tools/gemm_ksweep.py --flags 24=1 24=0 24=2 24=3 --iters 20 > $O/ksweep.log 2>&1
timeout 1200 python -m pytest tests/test_path_gpu.py tests/test_large_gpu.py tests/test_precision_gpu.py -m gpu -q -p no:cacheprovider > $O/tests_path.log 2>&1; tail -1 $O/tests_path.log
timeout 600 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-parity-mode > $O/bench.log 2>&1; tail -1 $O/bench.log | cut -c1-300
SAMAUDIO_DEBUG_FLAGS="24=1" timeout 600 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-parity-mode --no-roofline > $O/bench_general.log 2>&1; tail -1 $O/bench_general.log | cut -c1-300
timeout 600 python bench.py --steps 4 --warmup 2 --batch 4 --no-cpu-baseline --no-parity-mode --no-roofline > $O/bench_batch4.log 2>&1; tail -1 $O/bench_batch4.log | cut -c1-300
