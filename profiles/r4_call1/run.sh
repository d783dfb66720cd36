#!/bin/bash
# round 4, GPU call 1: the deep-pipeline form of the 8-phase kernel (gemm8d_kernel, debug flag 23) - hardware tests
# (bitwise vs gemm8 / gemm8s, 5 repetitions), time vs K at fixed (M, N) for both forms (per-K-tile slope and per-tile fixed
# cost), and the DiT shapes beside hipBLASLt with the flag A/B'd in one process.
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r4_call1
O=gpurun_out/r4_call1
export SAMAUDIO_DEBUG_FLAGS="23=1"
for i in 1 2 3 4 5; do
  timeout 600 python -m pytest tests/test_gemm2_gpu.py -m gpu -q -x -p no:cacheprovider -k "(22 and (plain_tails or asymmetric or gate_residual or conv_forms)) or 8phase_family or pipelined_form or tail_split" > $O/tests_deep_$i.log 2>&1
  tail -1 $O/tests_deep_$i.log
done
unset SAMAUDIO_DEBUG_FLAGS
timeout 600 python tools/gemm_ksweep.py --flags 23=0 23=1 --iters 20 > $O/ksweep.log 2>&1
timeout 900 python tools/gemm_bench.py --clips 16 32 4 --iters 20 --ab 23 > $O/gemm_bench_ab.log 2>&1
tail -5 $O/ksweep.log
