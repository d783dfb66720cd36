#!/bin/bash
# Round 2, GPU call 19: T5 prompt encoder on the HIP library (parity vs transformers' T5EncoderModel at t5-base dims, fp32 /
# bf16 / fp16), the all-layer cross-attention fold (tests + bench A/B against per-layer launches, flag 0), conv7h at
# C = 64 / 128 against the implicit GEMMs, bench --t5.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r2_call19
mkdir -p $OUT
(timeout 900 python -m pytest tests/test_t5_gpu.py tests/test_zz_next_rows_gpu.py tests/test_path_gpu.py tests/test_large_gpu.py -m gpu -q -s) > $OUT/gpu_tests_subset.log 2>&1; tail -2 $OUT/gpu_tests_subset.log; grep -E "^t5|t5-base|t5 small" $OUT/gpu_tests_subset.log
(timeout 400 python tools/op_bench.py) > $OUT/op_bench.log 2>&1; grep -E "conv7 C=(64|128)|cross_attn_fold" $OUT/op_bench.log
b() { name=$1; shift; (timeout 400 python bench.py --no-cpu-baseline "$@") > $OUT/bench_$name.log 2>&1; tail -1 $OUT/bench_$name.log | cut -c52-100; }
b new
SAMAUDIO_DEBUG_FLAGS=0=1 b fold_per_layer --no-roofline
b new_again --no-roofline
SAMAUDIO_DEBUG_FLAGS=0=1 b fold_per_layer_again --no-roofline
b t5 --t5 --no-roofline
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r2_call19/bench_new.log") if l.startswith("{")][-1])
for k in d["kernels"]:
    if "fold" in k["kernel"] or "probs" in k["kernel"]: print(k)
PY
