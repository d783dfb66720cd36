set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${R6_OUT:-r6_call20}
mkdir -p $O
python -m pytest tests/test_x3_gpu.py tests/test_hostile_gpu.py -m gpu -q -s -p no:cacheprovider > $O/tests_x3.log 2>&1
grep "x3 GEMM\|hostile\|passed\|failed" $O/tests_x3.log | cut -c1-260
Q="--no-cpu-baseline --no-parity-mode --no-other-configs --no-hostile"
show() { python - "$1" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print(sys.argv[1], d["value"], d["ms_per_step"], d.get("parity_check"))
for k in sorted(d["kernels"], key=lambda k: -k["ms"]):
    if k["ms"] > 20 and not k["kernel"].startswith("codec/"): print("   ", k["kernel"], k["launches"], k["ms"], k["tflops"], k["gbs"])
PY
}
timeout 500 python bench.py $Q --verify --steps 4 --warmup 1 > $O/bench_x3_b32_shared.log 2>&1; show $O/bench_x3_b32_shared.log
timeout 300 python bench.py $Q --no-verify --no-roofline --steps 3 --warmup 1 --batch 4 > $O/bench_x3_b4_shared.log 2>&1; tail -1 $O/bench_x3_b4_shared.log | cut -c1-160
