#!/bin/bash
# Round 2, GPU call 14: BK-32 multi-workgroup tiles for the 96 - 192-channel DAC stages (flag 15 = previous tiles): bench A/B on
# configs[2] and the codec-heavy configs[3]; GPU tests.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r2_call14
mkdir -p $OUT
(timeout 900 python -m pytest tests -m gpu -q) > $OUT/gpu_tests.log 2>&1; tail -2 $OUT/gpu_tests.log
b() { name=$1; shift; (timeout 400 python bench.py --no-cpu-baseline "$@") > $OUT/bench_$name.log 2>&1; tail -1 $OUT/bench_$name.log | cut -c52-100; }
b new
SAMAUDIO_DEBUG_FLAGS=15=1 b old --no-roofline
b new_again --no-roofline
SAMAUDIO_DEBUG_FLAGS=15=1 b old_again --no-roofline
b rerank_new --batch 8 --candidates 8 --predict-spans --steps 2 --no-roofline
SAMAUDIO_DEBUG_FLAGS=15=1 b rerank_old --batch 8 --candidates 8 --predict-spans --steps 2 --no-roofline
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r2_call14/bench_new.log") if l.startswith("{")][-1])
for k in d["kernels"]:
    if k["kernel"].startswith("codec"): print(k)
print(d["roofline_hbm"][0])
PY
