#!/bin/bash
# First GPU call of round 2 (run on the GPU box via gpurun, ~15 GPU-minutes):
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/round2_first_call.sh'
# 1. the whole -m gpu suite (incl. tests/test_zz_next_rows_gpu.py, the Judge / PE-AV / PE-A-Frame rows that round 1 could
#    only verify on the emulation / SIMT simulator) and the experimental GEMM variants' parity tests;
# 2. every experimental GEMM variant (loader-wave gemm5 family, BK-32 two-workgroup tile, 8-phase gemm8) timed next to
#    the shipped kernels at the guide's 4096^3 / 8192^3 and at the model's shapes;
# 3. the per-CU pipe probe (MFMA / LDS-fill / fragment-read interference: same wave, SIMD partner, dedicated waves);
# 4. the default bench line and the BASELINE configs[3] workload (reranking with the HIP Judge).
# Everything lands in gpurun_out/r2_first/.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r2_first
mkdir -p $OUT
(timeout 600 python -m pytest tests -m gpu -q -x) > $OUT/gpu_tests.log 2>&1; tail -3 $OUT/gpu_tests.log
(SAMAUDIO_TEST_EXPERIMENTAL=1 timeout 300 python -m pytest tests/test_gemm2_gpu.py -m gpu -q) > $OUT/gpu_tests_experimental.log 2>&1; tail -2 $OUT/gpu_tests_experimental.log
(timeout 300 python tools/gemm_bench.py --experimental) > $OUT/gemm_experimental.log 2>&1; tail -12 $OUT/gemm_experimental.log | cut -c1-600
(timeout 200 python tools/op_bench.py) > $OUT/op_bench.log 2>&1; tail -14 $OUT/op_bench.log   # incl. the two flag-gated candidates
if [ -x tools/cu_probe ]; then (timeout 120 tools/cu_probe) > $OUT/cu_probe.log 2>&1; tail -20 $OUT/cu_probe.log; fi
(timeout 400 python bench.py) > $OUT/bench.log 2>&1; tail -1 $OUT/bench.log | cut -c1-400
(timeout 400 python bench.py --batch 8 --candidates 8 --no-cpu-baseline --steps 2) > $OUT/bench_rerank.log 2>&1; tail -1 $OUT/bench_rerank.log | cut -c1-400
ls -la $OUT
