#!/bin/bash
# Round 5, GPU call 3: the shipped few-row defaults (K-tile-major weights, wave roles with PROD 2, prefetch chain without the fold GEMM)
# against their parts, per GEMM class and end to end; small* 8 clips and large* 8 clips too.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5_call3; mkdir -p $O
export OMP_NUM_THREADS=16
Q="--no-cpu-baseline --no-parity-mode --no-other-configs --no-verify"
run() {
  local name=$1; shift
  local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  ( env "${envs[@]}" timeout 300 python bench.py $Q "$@" ) > $O/$name.log 2> $O/$name.err
  python - "$O/$name.log" "$name" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(f"{sys.argv[2]:28s} {d['value']:8.2f} s-audio/s  {d['ms_per_step']:8.2f} ms")
    for k in d["kernels"]:
        if "#" in k["kernel"] or "--all" in sys.argv:
            print(f"    {k['kernel']:44s} {k['launches']:5d} {1e3*k['ms']/k['launches']:8.1f} us  {k['tflops']:7.1f} TF/s")
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
B4="--batch 4 --steps 6 --warmup 2"
# shipped defaults now: K-tile-major weights, wave roles (PROD 2), prefetch chain without the fold GEMM
run b4_default_cls       SAMAUDIO_PROF_BY_CLASS=1 -- $B4
run b4_nopf_cls          SAMAUDIO_PROF_BY_CLASS=1 SAMAUDIO_PREFETCH_ROWS=0 -- $B4
run b4_rows_pf_cls       SAMAUDIO_PROF_BY_CLASS=1 SAMAUDIO_WEIGHT_LAYOUT=rows -- $B4
run b4_default           -- $B4
run b4_nopf              SAMAUDIO_PREFETCH_ROWS=0 -- $B4
run b4_rows_pf           SAMAUDIO_WEIGHT_LAYOUT=rows -- $B4
run b4_rows_nopf_noroles SAMAUDIO_WEIGHT_LAYOUT=rows SAMAUDIO_PREFETCH_ROWS=0 SAMAUDIO_DEBUG_FLAGS=27=1 -- $B4
run b4_default_again     -- $B4
S8="--size small* --batch 8 --steps 6 --warmup 2"
run s8_default           -- $S8
run s8_nopf              SAMAUDIO_PREFETCH_ROWS=0 -- $S8
run s8_old               SAMAUDIO_WEIGHT_LAYOUT=rows SAMAUDIO_PREFETCH_ROWS=0 SAMAUDIO_DEBUG_FLAGS=27=1 -- $S8
run b8_default           -- --batch 8 --steps 4 --warmup 2
run b8_pf4096            SAMAUDIO_PREFETCH_ROWS=4096 -- --batch 8 --steps 4 --warmup 2
run b8_old               SAMAUDIO_WEIGHT_LAYOUT=rows SAMAUDIO_PREFETCH_ROWS=0 SAMAUDIO_DEBUG_FLAGS=27=1 -- --batch 8 --steps 4 --warmup 2
