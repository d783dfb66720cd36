#!/bin/bash
# round 4, GPU call 2: the linear epilogue of the 8-phase family (gemm8.hip epilogue8_linear; debug flag 24 = 1 -> the general
# LDS-staged epilogue).  Hardware tests (bitwise vs the general epilogue), time vs K for both (per-tile fixed cost), the DiT
# shapes beside hipBLASLt, the end-to-end parity tests, and the quick bench line A/B.
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r4_call2; mkdir -p $O
timeout 900 python -m pytest tests/test_gemm2_gpu.py tests/test_gemm_gpu.py -m gpu -q -x -p no:cacheprovider > $O/tests_gemm.log 2>&1; tail -1 $O/tests_gemm.log
timeout 600 python tools/gemm_ksweep.py --flags 24=1 24=0 --iters 20 > $O/ksweep.log 2>&1
timeout 900 python tools/gemm_bench.py --clips 16 4 --iters 20 --ab 24 > $O/gemm_bench_ab.log 2>&1
timeout 1200 python -m pytest tests/test_path_gpu.py tests/test_large_gpu.py tests/test_precision_gpu.py -m gpu -q -x -p no:cacheprovider > $O/tests_path.log 2>&1; tail -1 $O/tests_path.log
timeout 600 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-parity-mode > $O/bench_linear.log 2>&1; tail -1 $O/bench_linear.log | cut -c1-300
SAMAUDIO_DEBUG_FLAGS="24=1" timeout 600 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-parity-mode --no-roofline > $O/bench_general.log 2>&1; tail -1 $O/bench_general.log | cut -c1-300
timeout 600 python bench.py --steps 4 --warmup 2 --batch 4 --no-cpu-baseline --no-parity-mode --no-roofline > $O/bench_batch4.log 2>&1; tail -1 $O/bench_batch4.log | cut -c1-300
