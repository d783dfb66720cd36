#!/bin/bash
# Round 2, GPU call 24: can a streaming kernel run BESIDE the other row group's 8-phase GEMM?  gemm8 holds 2 x 232 of the
# 512 VGPRs of every SIMD; the two-pass rmsnorm_mod kernel (flag 2) needs 46 and fits into the rest, the register-resident
# one (102, the default: faster alone) does not.  Two-stream bench A/B, and the same with one stream as the control.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r2_call24
mkdir -p $OUT
b() { name=$1; shift; (timeout 400 python bench.py --no-cpu-baseline --no-roofline "$@") > $OUT/bench_$name.log 2>&1; echo "$name $(tail -1 $OUT/bench_$name.log | cut -c52-100)"; }
b default
SAMAUDIO_DEBUG_FLAGS=2=1 b twopass_rmsnorm
b default_again
SAMAUDIO_DEBUG_FLAGS=2=1 b twopass_rmsnorm_again
b s1_default --streams 1
SAMAUDIO_DEBUG_FLAGS=2=1 b s1_twopass --streams 1
