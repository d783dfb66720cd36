#!/bin/bash
# Round 2, GPU call 18: conv7h (k7 convolution with the activation halo tile resident in LDS) - bitwise test against the
# implicit GEMM on hardware, microbenchmarks at C = 96 / 192, bench A/B (flag 11 = implicit GEMMs) on configs[2] and [3].
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r2_call18
mkdir -p $OUT
(timeout 600 python -m pytest tests/test_gemm2_gpu.py tests/test_path_gpu.py tests/test_zz_next_rows_gpu.py -m gpu -q) > $OUT/gpu_tests_subset.log 2>&1; tail -2 $OUT/gpu_tests_subset.log
(timeout 400 python tools/op_bench.py) > $OUT/op_bench.log 2>&1; grep -E "conv7" $OUT/op_bench.log
b() { name=$1; shift; (timeout 400 python bench.py --no-cpu-baseline "$@") > $OUT/bench_$name.log 2>&1; tail -1 $OUT/bench_$name.log | cut -c52-100; }
b new
SAMAUDIO_DEBUG_FLAGS=11=1 b old --no-roofline
b new_again --no-roofline
SAMAUDIO_DEBUG_FLAGS=11=1 b old_again --no-roofline
b rerank_new --batch 8 --candidates 8 --predict-spans --steps 2 --no-roofline
SAMAUDIO_DEBUG_FLAGS=11=1 b rerank_old --batch 8 --candidates 8 --predict-spans --steps 2 --no-roofline
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r2_call18/bench_new.log") if l.startswith("{")][-1])
for k in d["kernels"]:
    if k["kernel"].startswith("codec"): print(k)
print(d["roofline_hbm"][0])
PY
