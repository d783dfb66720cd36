#!/bin/bash
# Round 5, GPU call 13: the launches of a solve replayed as a HIP graph (samaudio.h SAMAUDIO_OPT_ODE_GRAPH, environment
# SAMAUDIO_ODE_GRAPH=1 for SAMAudio's default): the bitwise test first, then A/B at the few-row shapes and at the headline.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5_call13; mkdir -p $O
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
    print(f"{sys.argv[2]:18s} {d['value']:8.2f} s-audio/s {d['ms_per_step']:8.2f} ms replays {d['config'].get('ode_graph_replays')} | gemm8 {k('dit/gemm8_bf16_256x256_8phase')} gemm8s {k('dit/gemm8s_bf16_128x128')} tail {k('dit/gemm8s_bf16_128x128_tail')}")
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
( SAMAUDIO_GRAPH_VERBOSE=1 timeout 600 python -m pytest tests/test_path_gpu.py -m gpu -q -s -k "ode_graph" ) > $O/test_graph.log 2>&1; echo "graph test exit=$?"; tail -3 $O/test_graph.log
run b32_eager     SAMAUDIO_ODE_GRAPH=0 -- --steps 5 --warmup 3
run b32_graph     SAMAUDIO_ODE_GRAPH=1 -- --steps 5 --warmup 3
tail -4 $O/b32_graph.err
run b4_eager      SAMAUDIO_ODE_GRAPH=0 -- --batch 4 --steps 8 --warmup 3
run b4_graph      SAMAUDIO_ODE_GRAPH=1 -- --batch 4 --steps 8 --warmup 3
