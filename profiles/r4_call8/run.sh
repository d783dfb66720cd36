#!/bin/bash
# round 4, GPU call 8: q/k-norm + RoPE + head split + V transpose as the qkv GEMM's epilogue (gemm8.hip epilogue8_qkv; debug
# flag 27 = 1: the separate qkv_prep pass).  Hardware tests, the end-to-end parity tests, quick bench lines A/B (twice each).
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r4_call8; mkdir -p $O
timeout 900 python -m pytest tests/test_gemm2_gpu.py tests/test_path_gpu.py tests/test_large_gpu.py tests/test_precision_gpu.py tests/test_fp16_gpu.py -m gpu -q -p no:cacheprovider > $O/tests.log 2>&1; tail -2 $O/tests.log
grep "large\* full solve\|large\* forward\|2-step" $O/tests.log | head -12
for f in 0 1 0 1; do SAMAUDIO_DEBUG_FLAGS="27=$f" timeout 600 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-parity-mode --no-roofline --no-other-configs > $O/bench_f${f}_$RANDOM.log 2>&1; echo "flag27=$f: $(tail -1 $O/bench_f${f}_*.log | grep -o '"value": [0-9.]*' | tail -1)"; done
