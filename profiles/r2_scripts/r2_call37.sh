#!/bin/bash
# Round 2, GPU call 37 (NEGATIVE RESULT, the kernel form was removed again - check out commit "gemm8s: 256x128 pipelined
# tile" to reproduce): 256x128 pipelined tile of gemm8s for few-row launches (flag 25 = 128x128 only): the whole -m gpu
# suite on it, GEMM micro-benchmarks at M = 1000, bench A/B at 4 clips and on small* 8 clips, default line as the control.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r2_call37
mkdir -p $OUT
(timeout 900 python -m pytest tests -m gpu -q) > $OUT/gpu_tests.log 2>&1; tail -2 $OUT/gpu_tests.log
(timeout 200 python tools/gemm_bench.py --family --batch 4) > $OUT/gemm_family_m1000_new.log 2>&1; grep -E "M=1000 " $OUT/gemm_family_m1000_new.log | sed -E 's/ +/ /g' | cut -c1-120
(SAMAUDIO_DEBUG_FLAGS=25=1 timeout 200 python tools/gemm_bench.py --family --batch 4) > $OUT/gemm_family_m1000_128only.log 2>&1; grep -E "M=1000 " $OUT/gemm_family_m1000_128only.log | sed -E 's/ +/ /g' | cut -c1-120
b() { name=$1; shift; (timeout 400 python bench.py --no-cpu-baseline --no-roofline "$@") > $OUT/bench_$name.log 2>&1; echo "$name $(tail -1 $OUT/bench_$name.log | cut -c52-100)"; }
b batch4_new --batch 4 --steps 5
SAMAUDIO_DEBUG_FLAGS=25=1 b batch4_128only --batch 4 --steps 5
b small_new --size 'small*' --batch 8 --steps 5
SAMAUDIO_DEBUG_FLAGS=25=1 b small_128only --size 'small*' --batch 8 --steps 5
b default
