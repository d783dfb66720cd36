set -x
O=gpurun_out/${R6_OUT:-r6_call14}
mkdir -p $O
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_x3_gpu.py tests/test_gemm_gpu.py -m gpu -q -s -p no:cacheprovider -k "split_on_the_fly or codec_roundtrip or gemm_gpu" > $O/tests_fly.log 2>&1
grep "codec in\|passed\|failed\|Error" $O/tests_fly.log | cut -c1-260
timeout 600 python tools/fly_probe.py > $O/fly_probe.log 2>&1; cat $O/fly_probe.log | cut -c1-400
Q="--no-cpu-baseline --no-parity-mode --no-other-configs --no-hostile"
show() { python - "$1" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print(sys.argv[1], d["value"], d["ms_per_step"], (d.get("parity_check") or {}).get("waveform_err"))
for k in sorted(d["kernels"], key=lambda k: -k["ms"]):
    if (k["kernel"].startswith("codec/") or "gemm_f32" in k["kernel"]) and k["ms"] > 5: print("   ", k["kernel"], k["launches"], k["ms"], k["tflops"], k["gbs"])
PY
}
timeout 500 python bench.py $Q --steps 4 --warmup 1 > $O/bench_x3_b32.log 2>&1; show $O/bench_x3_b32.log
for v in ${R6_VARIANTS:-36=1 36=2}; do SAMAUDIO_DEBUG_FLAGS=$v timeout 500 python bench.py $Q --no-verify --steps 4 --warmup 1 > $O/bench_x3_b32_$v.log 2>&1; show $O/bench_x3_b32_$v.log; done
