#!/bin/bash
# Round 2, GPU call 23: provenance of the PRIMARY roofline entry - rocprofv3 kernel trace of the single-stream command with
# the tail split off (debug flag 10), i.e. exactly the launches bench.py's instrumented step brackets with HIP events - and
# the bench line the way the driver runs it (20 steps, 5 warm-up).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r2_call23
mkdir -p $OUT
(timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5) > $OUT/bench_driver_like.log 2>&1; tail -1 $OUT/bench_driver_like.log | cut -c1-220
(SAMAUDIO_DEBUG_FLAGS=10=1 timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python bench.py --steps 1 --warmup 1 --streams 1 --no-cpu-baseline --no-roofline) > $OUT/trace.log 2>&1; echo trace exit=$?
python tools/rocpd_stats.py $(find $OUT/trace -name '*results.db' | head -1) > $OUT/kernel_stats_streams1_nosplit.md 2>$OUT/kernel_stats.err; head -8 $OUT/kernel_stats_streams1_nosplit.md | cut -c1-160
rm -rf $OUT/trace
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r2_call23/bench_driver_like.log") if l.startswith("{")][-1])
r = d["roofline"]
print("bench: gemm8 avg_launch_us", r["avg_launch_us"], "launches/step", r["launches_per_step"], "frac", r["frac"], "value", d["value"])
n = ms = 0
for k in d["kernels"]:
    if k["kernel"].endswith("gemm8_bf16_256x256_8phase"): n += k["launches"]; ms += k["ms"]
print("HIP events over every launch of the symbol (dit + codec + prep):", n, round(1e3 * ms / n, 2), "us")
PY
