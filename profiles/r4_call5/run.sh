#!/bin/bash
# round 4, GPU call 5: gemm8p_kernel - the 8-phase loop with the DMA issued as inline assembly (scalar base + 32-bit lane
# offset; hipcc then counts the LDS fragment reads itself instead of lgkmcnt(0) before every MFMA cluster), debug flag 23 = 2;
# = 3: plus the W fragments Bs0 of the next K-tile read in P4 (8/4/8/4 reads per phase instead of 12/4/8/0).  Hardware tests
# (bitwise vs gemm8 / gemm8s, 3 repetitions each), time vs K, the DiT shapes beside hipBLASLt, quick bench lines.
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r4_call5; mkdir -p $O
for f in 2 3; do for i in 1 2 3; do
  SAMAUDIO_DEBUG_FLAGS="23=$f" timeout 600 python -m pytest tests/test_gemm2_gpu.py -m gpu -q -x -p no:cacheprovider -k "(22 and (plain_tails or asymmetric or gate_residual or conv_forms or linear_epilogue)) or 8phase_family or pipelined_form or tail_split" > $O/tests_f${f}_$i.log 2>&1; tail -1 $O/tests_f${f}_$i.log
done; done
timeout 600 python tools/gemm_ksweep.py --flags 23=0 23=2 23=3 --iters 20 --kinds plain --shapes 4000x2816 4096x4096 4000x8448 > $O/ksweep.log 2>&1; grep -v amdgpu $O/ksweep.log
timeout 900 python tools/gemm_bench.py --clips 16 --iters 20 --ab 23 --ab-value 3 > $O/gemm_bench_f3.log 2>&1
timeout 900 python tools/gemm_bench.py --clips 16 --iters 20 --ab 23 --ab-value 2 --no-blas > $O/gemm_bench_f2.log 2>&1
for f in 0 2 3; do SAMAUDIO_DEBUG_FLAGS="23=$f" timeout 600 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-parity-mode --no-roofline > $O/bench_f$f.log 2>&1; tail -1 $O/bench_f$f.log | cut -c1-200; done
