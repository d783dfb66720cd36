#!/bin/bash
# Round 5, GPU call 5: the driver's default bench command on the reworked line - T5 encoder inside every timed step, no blocking host
# syncs inside separate(), per-config parity checks (configs[3]: candidates + Judge; configs[4]: tower features + visual solve) - and
# the GPU tests of what changed (path, T5, next rows, configs).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5_call5; mkdir -p $O
export OMP_NUM_THREADS=16
( timeout 900 python -m pytest tests/test_path_gpu.py tests/test_t5_gpu.py tests/test_zz_next_rows_gpu.py tests/test_configs_gpu.py -m gpu -q -x -p no:cacheprovider ) > $O/tests.log 2>&1; echo "tests exit=$?"; tail -3 $O/tests.log
( time timeout 1500 python bench.py --gpus 1 --steps 10 --warmup 3 ) > $O/bench_default.log 2> $O/bench_default.err; echo "bench exit=$?"
tail -3 $O/bench_default.err | cut -c1-300
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r5_call5/bench_default.log") if l.startswith("{")][-1])
print("value", d["value"], "ms", d["ms_per_step"], "t5", d["config"]["text_encoder_in_step"] is not None)
print("roofline", {k: d["roofline"][k] for k in ("kernel", "achieved", "frac", "avg_launch_us")})
print("parity", {k: d["parity_check"][k] for k in ("within_tolerance", "ode_latent_err", "waveform_err")})
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
for o in d.get("other_configs", []):
    if "error" in o:
        print("ERR", o)
        continue
    pc = o["parity_check"] or {}
    print(o["config_name"], o["value"], o["wall_s"], "s |", {k: v for k, v in pc.items() if k in ("within_tolerance", "ode_latent_err", "waveform_err", "tower_feature_err", "tower_feature_min_cosine", "judge_score_err", "argmax_equal", "selected_waveform_err", "oracle_seconds", "oracle_pass", "video_term_effect_on_latent")})
PY
