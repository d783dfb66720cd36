set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${R6_OUT:-r6_call25}
mkdir -p $O
for f in 39=2 39=4; do SAMAUDIO_DEBUG_FLAGS=$f python -m pytest tests/test_x3_gpu.py -m gpu -q -s -p no:cacheprovider -k "sharing" 2>&1 | tail -1; done > $O/tests_x3_sharing_tail.log 2>&1; cat $O/tests_x3_sharing_tail.log
PROBE_ROWS=4000 python tools/x3_probe.py > $O/x3_probe_shipped.log 2>&1; cat $O/x3_probe_shipped.log | cut -c1-260
PROBE_ROWS=4000 SAMAUDIO_DEBUG_FLAGS=39=2 python tools/x3_probe.py > $O/x3_probe_tail2.log 2>&1; cat $O/x3_probe_tail2.log | cut -c1-260
PROBE_ROWS=4000 SAMAUDIO_DEBUG_FLAGS=39=4 python tools/x3_probe.py > $O/x3_probe_tail4.log 2>&1; cat $O/x3_probe_tail4.log | cut -c1-260
PROBE_ROWS=4000 python tools/x3_probe.py > $O/x3_probe_shipped2.log 2>&1; cat $O/x3_probe_shipped2.log | cut -c1-260
Q="--no-cpu-baseline --no-parity-mode --no-other-configs --no-hostile --no-verify"
show() { python - "$1" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print(sys.argv[1], d["value"], d["ms_per_step"])
for k in sorted(d["kernels"], key=lambda k: -k["ms"]):
    if k["ms"] > 100 and not k["kernel"].startswith("codec/"): print("   ", k["kernel"], k["launches"], k["ms"], k["tflops"], k["gbs"])
PY
}
SAMAUDIO_DEBUG_FLAGS=39=2 timeout 500 python bench.py $Q --steps 4 --warmup 1 > $O/bench_b32_tail2.log 2>&1; show $O/bench_b32_tail2.log
SAMAUDIO_DEBUG_FLAGS=39=4 timeout 500 python bench.py $Q --steps 4 --warmup 1 > $O/bench_b32_tail4.log 2>&1; show $O/bench_b32_tail4.log
timeout 500 python bench.py $Q --steps 4 --warmup 1 > $O/bench_b32_shipped.log 2>&1; show $O/bench_b32_shipped.log
