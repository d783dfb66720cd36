#!/bin/bash
# Round 2, GPU call 22 (NEGATIVE RESULT, the kernel change was reverted - see DESIGN.md 3.4; this script needs commit
# "gemm8s: 4-stage ring ..." checked out to reproduce): gemm8s with a 4-stage ring for launches of <= 256 workgroups (flag 21 = the double buffer) - bitwise
# tests on hardware, bench A/B at 4 clips (strong-scaling share), small* 8 clips (configs[1]), one row group with the tail
# split, the default line; threshold between gemm8s and the 256x256 kernel (flag 22).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r2_call22
mkdir -p $OUT
(timeout 900 python -m pytest tests/test_gemm2_gpu.py tests/test_path_gpu.py -m gpu -q) > $OUT/gpu_tests_subset.log 2>&1; tail -2 $OUT/gpu_tests_subset.log
b() { name=$1; shift; (timeout 400 python bench.py --no-cpu-baseline "$@") > $OUT/bench_$name.log 2>&1; echo "$name $(tail -1 $OUT/bench_$name.log | cut -c52-100)"; }
b batch4_new --batch 4 --steps 5
SAMAUDIO_DEBUG_FLAGS=21=1 b batch4_old --batch 4 --steps 5 --no-roofline
SAMAUDIO_DEBUG_FLAGS=22=256 b batch4_thr256 --batch 4 --steps 5 --no-roofline
SAMAUDIO_DEBUG_FLAGS=22=512 b batch4_thr512 --batch 4 --steps 5 --no-roofline
b small_new --size 'small*' --batch 8 --steps 5
SAMAUDIO_DEBUG_FLAGS=21=1 b small_old --size 'small*' --batch 8 --steps 5 --no-roofline
SAMAUDIO_DEBUG_FLAGS=22=256 b small_thr256 --size 'small*' --batch 8 --steps 5 --no-roofline
b streams1_new --streams 1
SAMAUDIO_DEBUG_FLAGS=21=1 b streams1_old --streams 1 --no-roofline
b default_new --no-roofline
SAMAUDIO_DEBUG_FLAGS=21=1 b default_old --no-roofline
python - <<'PY'
import json
for n in ("batch4_new", "small_new", "streams1_new"):
    d = json.loads([l for l in open(f"gpurun_out/r2_call22/bench_{n}.log") if l.startswith("{")][-1])
    print(n, d["value"])
    for k in d["kernels"]:
        if "gemm8" in k["kernel"] and k["kernel"].startswith("dit"): print("   ", k["kernel"], k["launches"], k["ms"], k["tflops"])
PY
