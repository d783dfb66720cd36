#!/bin/bash
# Round 2, GPU call 3: LDS-staged coalesced epilogues (gemm8 + the 32x32x16 family), M-aware tile policy, A/Bs.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r2_call3
mkdir -p $OUT
(SAMAUDIO_TEST_EXPERIMENTAL=1 timeout 300 python -m pytest tests/test_gemm2_gpu.py -m gpu -q -x) > $OUT/gemm2_tests.log 2>&1; tail -2 $OUT/gemm2_tests.log
(timeout 300 python tools/gemm_bench.py --experimental) > $OUT/gemm_lds_epilogue.log 2>&1; tail -7 $OUT/gemm_lds_epilogue.log | cut -c1-200
(SAMAUDIO_DEBUG_FLAGS=8=1 timeout 300 python tools/gemm_bench.py --experimental) > $OUT/gemm_direct_epilogue.log 2>&1; tail -7 $OUT/gemm_direct_epilogue.log | cut -c1-200
b() { name=$1; shift; (timeout 300 python bench.py --no-cpu-baseline "$@") > $OUT/bench_$name.log 2>&1; tail -1 $OUT/bench_$name.log | cut -c1-160; }
b default
SAMAUDIO_DEBUG_FLAGS=9=1 b nonpersist_ldsepi
SAMAUDIO_DEBUG_FLAGS=8=1 b direct_epilogue
SAMAUDIO_DEBUG_FLAGS=7=1 b gemm8_everywhere
SAMAUDIO_DEBUG_FLAGS=7=1 b gemm8_everywhere_s2 --streams 2 --no-roofline
SAMAUDIO_DEBUG_FLAGS=7=1 SAMAUDIO_ALLOW_STREAMS=1 b gemm8_everywhere_s4 --streams 4 --no-roofline
b default_s2 --streams 2 --no-roofline
SAMAUDIO_DEBUG_FLAGS=9=1 b nonpersist_s2 --streams 2 --no-roofline
b batch4 --batch 4 --steps 5
SAMAUDIO_DEBUG_FLAGS=6=1 b batch4_mblind --batch 4 --steps 5 --no-roofline
ls -la $OUT
