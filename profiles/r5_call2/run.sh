#!/bin/bash
# Round 5, GPU call 2: where the few-row GEMM time sits, per GEMM class (SAMAUDIO_PROF_BY_CLASS=1: one HIP-event record per class), and
# a few more scheduling variants at 4 clips: two row groups with the 256x256 kernel from 60 tiles on; roles 2 + prefetch; repeats.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5_call2; mkdir -p $O
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
run b4_rows_cls          SAMAUDIO_PROF_BY_CLASS=1 SAMAUDIO_WEIGHT_LAYOUT=rows SAMAUDIO_PREFETCH_ROWS=0 -- $B4
run b4_ktm_pf_roles2_cls SAMAUDIO_PROF_BY_CLASS=1 SAMAUDIO_WEIGHT_LAYOUT=ktm SAMAUDIO_PREFETCH_ROWS=2048 SAMAUDIO_DEBUG_FLAGS=27=3 -- $B4
run b4_ktm_pf_roles2     SAMAUDIO_WEIGHT_LAYOUT=ktm SAMAUDIO_PREFETCH_ROWS=2048 SAMAUDIO_DEBUG_FLAGS=27=3 -- $B4
run b4_ktm_roles2        SAMAUDIO_WEIGHT_LAYOUT=ktm SAMAUDIO_PREFETCH_ROWS=0 SAMAUDIO_DEBUG_FLAGS=27=3 -- $B4
run b4_rows              SAMAUDIO_WEIGHT_LAYOUT=rows SAMAUDIO_PREFETCH_ROWS=0 -- $B4
run b4_s2_t60            SAMAUDIO_WEIGHT_LAYOUT=ktm SAMAUDIO_PREFETCH_ROWS=0 SAMAUDIO_DEBUG_FLAGS=30=60 -- $B4 --streams 2
run b4_s2_t60_roles2     SAMAUDIO_WEIGHT_LAYOUT=ktm SAMAUDIO_PREFETCH_ROWS=0 SAMAUDIO_DEBUG_FLAGS=30=60,27=3 -- $B4 --streams 2
run b4_s2_t30            SAMAUDIO_WEIGHT_LAYOUT=ktm SAMAUDIO_PREFETCH_ROWS=0 SAMAUDIO_DEBUG_FLAGS=30=30 -- $B4 --streams 2
run b4_t30               SAMAUDIO_WEIGHT_LAYOUT=ktm SAMAUDIO_PREFETCH_ROWS=0 SAMAUDIO_DEBUG_FLAGS=30=30 -- $B4
run b8_rows_cls          SAMAUDIO_PROF_BY_CLASS=1 SAMAUDIO_WEIGHT_LAYOUT=rows SAMAUDIO_PREFETCH_ROWS=0 -- --batch 8 --steps 4 --warmup 2
run b32_rows_cls         SAMAUDIO_PROF_BY_CLASS=1 SAMAUDIO_WEIGHT_LAYOUT=rows SAMAUDIO_PREFETCH_ROWS=0 -- --steps 3 --warmup 1
# isolated: the same shapes alone on the GPU, beside hipBLASLt (row-major weights)
( timeout 300 python tools/gemm_bench.py --clips 4 --roles ) > $O/gemm_bench_m1000_roles.log 2>&1; tail -8 $O/gemm_bench_m1000_roles.log | cut -c1-400
( timeout 300 python tools/gemm_bench.py --clips 4 ) > $O/gemm_bench_m1000.log 2>&1; tail -8 $O/gemm_bench_m1000.log | cut -c1-400
