#!/bin/bash
# round 4, GPU call 7: gemm8_kernel as a persistent kernel (debug flag 26 = 1: at most one workgroup per CU walking its
# XCD's run of tiles; the 16-bit epilogue's stores drain under the next tile's prologue) - hardware tests, the DiT shapes
# with the flag A/B'd in one process, quick bench lines with / without it (two row groups: the launches of the two
# groups then hold the CUs for a whole launch instead of a tile).
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r4_call7; mkdir -p $O
timeout 600 python -m pytest tests/test_gemm2_gpu.py -m gpu -q -p no:cacheprovider -k "persistent or 8phase_family or tail_split or linear_epilogue" > $O/tests.log 2>&1; tail -1 $O/tests.log
SAMAUDIO_DEBUG_FLAGS="26=1" timeout 600 python -m pytest tests/test_gemm2_gpu.py tests/test_path_gpu.py -m gpu -q -p no:cacheprovider > $O/tests_flag26.log 2>&1; tail -1 $O/tests_flag26.log
timeout 900 python tools/gemm_bench.py --clips 16 32 --iters 20 --ab 26 --no-blas > $O/gemm_bench_ab26.log 2>&1
for f in 0 1 0 1; do SAMAUDIO_DEBUG_FLAGS="26=$f" timeout 600 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-parity-mode --no-roofline --no-other-configs > $O/bench_f${f}_$RANDOM.log 2>&1; echo "flag26=$f: $(tail -1 $O/bench_f${f}_*.log | grep -o '"value": [0-9.]*' | tail -1)"; done
