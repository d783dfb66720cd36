#!/bin/bash
# Round 5, GPU call 12: the floor of the tail split (debug flag 33; shipped 16 tiles).  At 8 clips per GPU (the 4-GPU share of the 32-clip
# batch, one row group) qkv is 264 tiles of 256x256 = one round of the chip + 8 tiles, i.e. a second round as long as the first.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5_call12; mkdir -p $O
export OMP_NUM_THREADS=16
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
    print(f"{sys.argv[2]:18s} {d['value']:8.2f} s-audio/s {d['ms_per_step']:8.2f} ms | gemm8 {k('dit/gemm8_bf16_256x256_8phase')} gemm8s {k('dit/gemm8s_bf16_128x128')} tail {k('dit/gemm8s_bf16_128x128_tail')}")
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
run b8_floor16    X=1 -- --batch 8 --steps 6 --warmup 2
run b8_floor8     SAMAUDIO_DEBUG_FLAGS=33=8 -- --batch 8 --steps 6 --warmup 2
run b8_floor4     SAMAUDIO_DEBUG_FLAGS=33=4 -- --batch 8 --steps 6 --warmup 2
run b8_floor16_2  X=1 -- --batch 8 --steps 6 --warmup 2
run b4_floor8     SAMAUDIO_DEBUG_FLAGS=33=8 -- --batch 4 --steps 6 --warmup 2
run b4_floor16    X=1 -- --batch 4 --steps 6 --warmup 2
run s8_floor8     SAMAUDIO_DEBUG_FLAGS=33=8 -- --size 'small*' --batch 8 --steps 6 --warmup 2
run s8_floor16    X=1 -- --size 'small*' --batch 8 --steps 6 --warmup 2
