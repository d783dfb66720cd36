#!/bin/bash
# Round 5, GPU call 11: the one-kernel self-attention, second form (V through a [key][d] LDS staging tile and column reads instead of
# a 2-byte scatter; k-norm weights in registers): is a row still batch-dependent (call 10: the first form was, the two kernels are not)?
# And its time against the two kernels, in the model.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5_call11; mkdir -p $O
export OMP_NUM_THREADS=16
t() { name=$1; shift; ( env "$@" timeout 600 python -m pytest tests/test_configs_gpu.py -m gpu -q -p no:cacheprovider -k "sixty_four and fp16" ) > $O/t_$name.log 2>&1; echo "$name: $(tail -1 $O/t_$name.log)"; }
t default X=1
t attn_two_kernels SAMAUDIO_DEBUG_FLAGS=32=1
Q="--no-cpu-baseline --no-parity-mode --no-other-configs --no-verify"
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
    print(f"{sys.argv[2]:18s} {d['value']:8.2f} s-audio/s {d['ms_per_step']:8.2f} ms | attn_qkv {k('dit/self_attention_qkv')} attn {k('dit/self_attention')} prep {k('dit/qkv_prep')}")
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
run b4_fused      X=1 -- --batch 4 --steps 8 --warmup 2
run b4_two        SAMAUDIO_DEBUG_FLAGS=32=1 -- --batch 4 --steps 8 --warmup 2
run b4_fused_2    X=1 -- --batch 4 --steps 8 --warmup 2
run b32_fused     X=1 -- --steps 6 --warmup 2
run b32_two       SAMAUDIO_DEBUG_FLAGS=32=1 -- --steps 6 --warmup 2
