#!/bin/bash
# Round 2, GPU call 21: 4-wave residual-unit kernel for 96 channels (two workgroups per CU), 8-phase family for the DAC
# stages with 256 - 768 channels (flag 19), each as bench A/B against the current defaults.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r2_call21
mkdir -p $OUT
(timeout 900 python -m pytest tests/test_gemm2_gpu.py tests/test_path_gpu.py -m gpu -q) > $OUT/gpu_tests_subset.log 2>&1; tail -2 $OUT/gpu_tests_subset.log
(timeout 500 python tools/op_bench.py) > $OUT/op_bench.log 2>&1; grep -E "residual unit" $OUT/op_bench.log
b() { name=$1; shift; (timeout 400 python bench.py --no-cpu-baseline "$@") > $OUT/bench_$name.log 2>&1; echo "$name $(tail -1 $OUT/bench_$name.log | cut -c52-100)"; }
b default
SAMAUDIO_DEBUG_FLAGS=20=1 b ru96_4waves --no-roofline
SAMAUDIO_DEBUG_FLAGS=19=1 b wide_codec_8phase
SAMAUDIO_DEBUG_FLAGS=19=2 b wide_codec_8phase_384_on_128 
b default_again --no-roofline
SAMAUDIO_DEBUG_FLAGS=19=1,20=1 b both --no-roofline
python - <<'PY'
import json
for n in ("default", "wide_codec_8phase", "wide_codec_8phase_384_on_128"):
    d = json.loads([l for l in open(f"gpurun_out/r2_call21/bench_{n}.log") if l.startswith("{")][-1])
    print(n, d["value"])
    for k in d["kernels"]:
        if k["kernel"].startswith("codec"): print("   ", k["kernel"], k["launches"], k["ms"], k["tflops"], k["gbs"])
PY
