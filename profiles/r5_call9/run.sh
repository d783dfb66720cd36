#!/bin/bash
# Round 5, GPU call 9: self-attention straight from the qkv rows (one kernel; debug flag 32 = 1: qkv_prep + self-attention as before)
# - its kernel test on hardware, the path / large-dims tests on it, and the A/B end to end at 4 / 32 clips and small* 8 clips.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5_call9; mkdir -p $O
export OMP_NUM_THREADS=16
( timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_path_gpu.py tests/test_precision_gpu.py tests/test_configs_gpu.py -m gpu -q -s -p no:cacheprovider ) > $O/tests.log 2>&1; echo "tests exit=$?"; grep "qkv rows\|one kernel vs\|passed\|failed" $O/tests.log | cut -c1-200
Q="--no-cpu-baseline --no-parity-mode --no-other-configs"
run() { name=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  ( env "${envs[@]}" timeout 400 python bench.py $Q "$@" ) > $O/$name.log 2> $O/$name.err
  python - "$O/$name.log" "$name" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    ks = {k["kernel"]: k for k in (d.get("kernels") or [])}
    def k(n):
        x = ks.get(n)
        return f"{x['ms']:.1f}ms/{x['launches']}" if x else "-"
    pc = d.get("parity_check") or {}
    print(f"{sys.argv[2]:18s} {d['value']:8.2f} s-audio/s {d['ms_per_step']:8.2f} ms | attn_qkv {k('dit/self_attention_qkv')} attn {k('dit/self_attention')} prep {k('dit/qkv_prep')} | parity lat {pc.get('ode_latent_err')} wav {pc.get('waveform_err')}")
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
run b4_fused      X=1 -- --batch 4 --steps 8 --warmup 2 --no-verify
run b4_two        SAMAUDIO_DEBUG_FLAGS=32=1 -- --batch 4 --steps 8 --warmup 2 --no-verify
run b4_fused_2    X=1 -- --batch 4 --steps 8 --warmup 2 --no-verify
run s8_fused      X=1 -- --size 'small*' --batch 8 --steps 8 --warmup 2 --no-verify
run s8_two        SAMAUDIO_DEBUG_FLAGS=32=1 -- --size 'small*' --batch 8 --steps 8 --warmup 2 --no-verify
run b32_fused     X=1 -- --steps 6 --warmup 2 --verify
run b32_two       SAMAUDIO_DEBUG_FLAGS=32=1 -- --steps 6 --warmup 2 --no-verify
run b32_fused_2   X=1 -- --steps 6 --warmup 2 --no-verify
