#!/bin/bash
# Round 2, GPU call 12: A/B by flags in one build - 16-wave self-attention (flag 13 = 8 waves), cross_attn_fold batch
# split (flag 12), 256x64 tile for the 64-channel convolutions (flag 14 = old 128x64 tile).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r2_call12
mkdir -p $OUT
(timeout 600 python -m pytest tests/test_gemm2_gpu.py tests/test_kernels_gpu.py tests/test_path_gpu.py tests/test_large_gpu.py -m gpu -q) > $OUT/gpu_tests_subset.log 2>&1; tail -2 $OUT/gpu_tests_subset.log
(timeout 300 python tools/op_bench.py) > $OUT/op_bench.log 2>&1; grep -E "self_attention|fold" $OUT/op_bench.log
b() { name=$1; shift; (timeout 400 python bench.py --no-cpu-baseline "$@") > $OUT/bench_$name.log 2>&1; tail -1 $OUT/bench_$name.log | cut -c52-100; }
b default
SAMAUDIO_DEBUG_FLAGS=13=1 b attn8 --no-roofline
SAMAUDIO_DEBUG_FLAGS=12=1 b fold1 --no-roofline
SAMAUDIO_DEBUG_FLAGS=14=1 b codec64_old --no-roofline
SAMAUDIO_DEBUG_FLAGS=12=1,13=1,14=1 b all_old --no-roofline
b default_again --no-roofline
python - <<'PY'
import json
for n in ("default", ):
    d = json.loads([l for l in open(f"gpurun_out/r2_call12/bench_{n}.log") if l.startswith("{")][-1])
    for k in d["kernels"][:14]:
        print(k)
PY
ls $OUT
