#!/bin/bash
# Round 2, GPU call 33: RMSNorm + modulate with the five operand vectors pre-combined into two per evaluation (flag 24 = the
# five-vector kernel): the whole -m gpu suite on it (fp32 rounding differs in the last bit), bench A/B, per-kernel times.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r2_call33
mkdir -p $OUT
(timeout 900 python -m pytest tests -m gpu -q) > $OUT/gpu_tests.log 2>&1; tail -2 $OUT/gpu_tests.log
b() { name=$1; shift; (timeout 400 python bench.py --no-cpu-baseline "$@") > $OUT/bench_$name.log 2>&1; echo "$name $(tail -1 $OUT/bench_$name.log | cut -c52-100)"; }
b new
SAMAUDIO_DEBUG_FLAGS=24=1 b old
b new_again --no-roofline
SAMAUDIO_DEBUG_FLAGS=24=1 b old_again --no-roofline
python - <<'PY'
import json
for n in ("new", "old"):
    d = json.loads([l for l in open(f"gpurun_out/r2_call33/bench_{n}.log") if l.startswith("{")][-1])
    for k in d["kernels"]:
        if "rmsnorm" in k["kernel"] or "mod_tables" in k["kernel"]: print(n, k["kernel"], k["launches"], k["ms"], k.get("gbs"))
PY
