#!/bin/bash
# Round 5, GPU call 1: the few-row GEMM path.  (a) the new bitwise tests on hardware (K-tile-major weights, prefetch workgroups, wave
# roles; model-level layout / prefetch invariance), (b) A/B of the candidates INSIDE THE MODEL on one box: 4 clips (one GPU's share of
# the 32-clip batch over 8 GPUs), small* 8 clips, and the 32-clip headline for regression.
# Q = quick bench line (no CPU oracle, no side-by-side mode, no other configs).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5_call1; mkdir -p $O
export OMP_NUM_THREADS=16
( timeout 600 python -m pytest tests/test_gemm2_gpu.py -m gpu -q -x -p no:cacheprovider -k "ktm or pipelined_form or wave_roles or tail_split" ) > $O/tests_gemm.log 2>&1; echo "gemm tests exit=$?"; tail -2 $O/tests_gemm.log
( timeout 600 python -m pytest tests/test_path_gpu.py tests/test_fp16_gpu.py -m gpu -q -x -p no:cacheprovider -k "weight_layout or mixed_mode or sharding or concurrent" ) > $O/tests_path.log 2>&1; echo "path tests exit=$?"; tail -2 $O/tests_path.log
Q="--no-cpu-baseline --no-parity-mode --no-other-configs --no-verify"
run() {  # name, env..., -- bench args
  local name=$1; shift
  local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  ( env "${envs[@]}" timeout 300 python bench.py $Q "$@" ) > $O/$name.log 2> $O/$name.err
  python - "$O/$name.log" "$name" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    r = d.get("roofline") or {}
    ks = {k["kernel"]: k for k in (d.get("kernels") or [])}
    def k(n):
        x = ks.get(n)
        return f"{x['ms']:.1f}ms/{x['launches']}" if x else "-"
    print(f"{sys.argv[2]:28s} {d['value']:8.2f} s-audio/s  {d['ms_per_step']:8.2f} ms  dom {r.get('kernel')} {r.get('avg_launch_us')} us frac {r.get('frac')}  gemm8s {k('dit/gemm8s_bf16_128x128')} gemm8 {k('dit/gemm8_bf16_256x256_8phase')}")
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
B4="--batch 4 --steps 6 --warmup 2"
run b4_rows               SAMAUDIO_WEIGHT_LAYOUT=rows SAMAUDIO_PREFETCH_ROWS=0 -- $B4
run b4_ktm                SAMAUDIO_WEIGHT_LAYOUT=ktm  SAMAUDIO_PREFETCH_ROWS=0 -- $B4
run b4_ktm_roles0         SAMAUDIO_WEIGHT_LAYOUT=ktm  SAMAUDIO_PREFETCH_ROWS=0 SAMAUDIO_DEBUG_FLAGS=27=2 -- $B4
run b4_ktm_roles2         SAMAUDIO_WEIGHT_LAYOUT=ktm  SAMAUDIO_PREFETCH_ROWS=0 SAMAUDIO_DEBUG_FLAGS=27=3 -- $B4
run b4_ktm_pf             SAMAUDIO_WEIGHT_LAYOUT=ktm  SAMAUDIO_PREFETCH_ROWS=2048 -- $B4
run b4_ktm_pf_roles0      SAMAUDIO_WEIGHT_LAYOUT=ktm  SAMAUDIO_PREFETCH_ROWS=2048 SAMAUDIO_DEBUG_FLAGS=27=2 -- $B4
run b4_rows_pf_roles0     SAMAUDIO_WEIGHT_LAYOUT=rows SAMAUDIO_PREFETCH_ROWS=2048 SAMAUDIO_DEBUG_FLAGS=27=2 -- $B4
run b4_rows_roles0        SAMAUDIO_WEIGHT_LAYOUT=rows SAMAUDIO_PREFETCH_ROWS=0 SAMAUDIO_DEBUG_FLAGS=27=2 -- $B4
run b4_ktm_pf_roles0_t200 SAMAUDIO_WEIGHT_LAYOUT=ktm  SAMAUDIO_PREFETCH_ROWS=2048 SAMAUDIO_DEBUG_FLAGS=27=2,30=200 -- $B4
run b4_ktm_pf_roles0_s2   SAMAUDIO_WEIGHT_LAYOUT=ktm  SAMAUDIO_PREFETCH_ROWS=2048 SAMAUDIO_DEBUG_FLAGS=27=2 -- $B4 --streams 2
run b4_rows_again         SAMAUDIO_WEIGHT_LAYOUT=rows SAMAUDIO_PREFETCH_ROWS=0 -- $B4
S8="--size small* --batch 8 --steps 6 --warmup 2"
run s8_rows               SAMAUDIO_WEIGHT_LAYOUT=rows SAMAUDIO_PREFETCH_ROWS=0 -- $S8
run s8_ktm                SAMAUDIO_WEIGHT_LAYOUT=ktm  SAMAUDIO_PREFETCH_ROWS=0 -- $S8
run s8_ktm_pf_roles0      SAMAUDIO_WEIGHT_LAYOUT=ktm  SAMAUDIO_PREFETCH_ROWS=2048 SAMAUDIO_DEBUG_FLAGS=27=2 -- $S8
run s8_ktm_pf             SAMAUDIO_WEIGHT_LAYOUT=ktm  SAMAUDIO_PREFETCH_ROWS=2048 -- $S8
B32="--steps 4 --warmup 2"
run b32_rows              SAMAUDIO_WEIGHT_LAYOUT=rows SAMAUDIO_PREFETCH_ROWS=0 -- $B32
run b32_ktm               SAMAUDIO_WEIGHT_LAYOUT=ktm  SAMAUDIO_PREFETCH_ROWS=0 -- $B32
run b32_ktm_pf            SAMAUDIO_WEIGHT_LAYOUT=ktm  SAMAUDIO_PREFETCH_ROWS=100000 -- $B32
