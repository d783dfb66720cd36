#!/bin/bash
# round 4, GPU call 12: precision="mixed" - bfloat16 operands on the five big GEMM classes of the DiT layers inside the fp16
# build (SAMAUDIO_OPT_ALT16_CLASSES), fp16 elsewhere.  Kernel tests, the parity tests at large* / small*, bench lines of the
# three 16-bit modes on one box (each with its own parity_check).
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r4_call12; mkdir -p $O
timeout 600 python -m pytest tests/test_fp16_gpu.py tests/test_gemm2_gpu.py -m gpu -q -p no:cacheprovider -s -k "mixed or fp16 or 8phase or linear_epilogue" > $O/tests_kernels.log 2>&1; tail -1 $O/tests_kernels.log; grep "mini separate" $O/tests_kernels.log
timeout 1200 python -m pytest tests/test_large_gpu.py tests/test_configs_gpu.py -m gpu -q -p no:cacheprovider -s -k "mixed or fp16" > $O/tests_parity.log 2>&1; tail -1 $O/tests_parity.log; grep "mixed\|fp16" $O/tests_parity.log | grep -i "err\|solve" | head -12
for prec in mixed fp16 bf16 mixed; do timeout 900 python bench.py --precision $prec --steps 6 --warmup 2 --no-cpu-baseline --verify --no-parity-mode --no-other-configs > $O/bench_${prec}_$RANDOM.log 2>&1; f=$(ls -t $O/bench_${prec}_*.log | head -1); echo "$prec: $(grep -o '"value": [0-9.]*' $f | head -1) $(grep -o '"frac": [0-9.]*' $f | head -1) $(grep -o '"within_tolerance": [a-z]*' $f | head -1) $(grep -o '"ode_latent_err": [0-9.e-]*' $f | head -1) $(grep -o '"waveform_err": [0-9.e-]*' $f | head -1)"; done
