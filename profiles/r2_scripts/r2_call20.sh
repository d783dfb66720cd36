#!/bin/bash
# Round 2, GPU call 20: fused DAC residual units (bitwise test on hardware, microbenchmarks per channel count, bench A/B per
# channel set), cross_attn_fold with the XCD-major workgroup deal (microbenchmark + bench A/B), 3 concurrent row groups.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r2_call20
mkdir -p $OUT
(timeout 900 python -m pytest tests/test_gemm2_gpu.py tests/test_kernels_gpu.py tests/test_path_gpu.py tests/test_large_gpu.py -m gpu -q) > $OUT/gpu_tests_subset.log 2>&1; tail -2 $OUT/gpu_tests_subset.log
(timeout 500 python tools/op_bench.py) > $OUT/op_bench.log 2>&1; grep -E "residual unit|cross_attn_fold  " $OUT/op_bench.log
(SAMAUDIO_DEBUG_FLAGS=0=1 timeout 300 python tools/op_bench.py) 2>&1 | grep -E "^cross_attn_fold  " | sed 's/^/heads-on-blockIdx.y: /' | tee $OUT/op_bench_fold_old.log
b() { name=$1; shift; (timeout 400 python bench.py --no-cpu-baseline "$@") > $OUT/bench_$name.log 2>&1; echo "$name $(tail -1 $OUT/bench_$name.log | cut -c52-100)"; }
b new
SAMAUDIO_DEBUG_FLAGS=16=1 b no_resunit --no-roofline
SAMAUDIO_DEBUG_FLAGS=0=1 b old_fold --no-roofline
SAMAUDIO_DEBUG_FLAGS=17=12 b resunit_64_96_only --no-roofline
b new_again --no-roofline
SAMAUDIO_DEBUG_FLAGS=16=1,0=1 b neither --no-roofline
SAMAUDIO_ALLOW_STREAMS=1 b streams3 --streams 3 --no-roofline
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r2_call20/bench_new.log") if l.startswith("{")][-1])
for k in d["kernels"]:
    if k["kernel"].startswith("codec") or "fold" in k["kernel"]: print(k)
print(d["roofline_hbm"][0])
PY
