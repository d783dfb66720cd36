#!/bin/bash
# Round 2, GPU call 25: gemm8s in its pipelined form (3-stage ring, fragments of K-tile t+1 read underneath the MFMAs of
# K-tile t) for launches of <= 256 workgroups (flag 21 = plain form) - bitwise tests on hardware, gemm micro-benchmarks at
# M = 1000, bench A/B at 4 clips (strong-scaling share), small* 8 clips (configs[1]), one row group with the tail split.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r2_call25
mkdir -p $OUT
(timeout 900 python -m pytest tests/test_gemm2_gpu.py tests/test_path_gpu.py -m gpu -q) > $OUT/gpu_tests_subset.log 2>&1; tail -2 $OUT/gpu_tests_subset.log
(timeout 300 python tools/gemm_bench.py --family --batch 4) > $OUT/gemm_family_m1000_new.log 2>&1; grep -E "M=1000 " $OUT/gemm_family_m1000_new.log | cut -c1-230
(SAMAUDIO_DEBUG_FLAGS=21=1 timeout 300 python tools/gemm_bench.py --family --batch 4) > $OUT/gemm_family_m1000_plain.log 2>&1; grep -E "M=1000 " $OUT/gemm_family_m1000_plain.log | cut -c1-230
b() { name=$1; shift; (timeout 400 python bench.py --no-cpu-baseline "$@") > $OUT/bench_$name.log 2>&1; echo "$name $(tail -1 $OUT/bench_$name.log | cut -c52-100)"; }
b batch4_new --batch 4 --steps 5
SAMAUDIO_DEBUG_FLAGS=21=1 b batch4_plain --batch 4 --steps 5 --no-roofline
b small_new --size 'small*' --batch 8 --steps 5
SAMAUDIO_DEBUG_FLAGS=21=1 b small_plain --size 'small*' --batch 8 --steps 5 --no-roofline
b streams1_new --streams 1
SAMAUDIO_DEBUG_FLAGS=21=1 b streams1_plain --streams 1 --no-roofline
b default_new --no-roofline
python - <<'PY'
import json
for n in ("batch4_new", "small_new", "streams1_new"):
    d = json.loads([l for l in open(f"gpurun_out/r2_call25/bench_{n}.log") if l.startswith("{")][-1])
    print(n, d["value"])
    for k in d["kernels"]:
        if "gemm8" in k["kernel"] and k["kernel"].startswith("dit"): print("   ", k["kernel"], k["launches"], k["ms"], k["tflops"])
PY
