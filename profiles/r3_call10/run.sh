#!/bin/bash
# GPU call 10 of round 3: two re-orderings of the 8-phase K loop, A/B in one library (debug flag 22): 1 = the staging
# instructions of a phase issue before its ds_reads; 2 = two super-phases per K-tile (32-MFMA clusters, 4 barriers instead of 8).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3_call10
mkdir -p $O
for f in 0 1 2; do
  ( SAMAUDIO_DEBUG_FLAGS=22=$f timeout 300 python tools/gemm_bench.py --clips 32 16 --iters 20 --no-blas ) > $O/gemm_bench_flag22_$f.log 2>&1
done
Q="--no-cpu-baseline --no-parity-mode --steps 4 --warmup 1"
for f in 0 1 2; do
  ( SAMAUDIO_DEBUG_FLAGS=22=$f timeout 300 python bench.py $Q ) > $O/bench_flag22_$f.log 2>&1
done
for f in 0 1 2; do echo "flag22=$f"; grep -o '"value": [0-9.]*' $O/bench_flag22_$f.log | head -1; grep -v amdgpu $O/gemm_bench_flag22_$f.log | sed 's/ok  *([^)]*)/ok/g' | grep "M=8000\|M=4000\|4096" | awk -F'|' '{print $1, $3}' | cut -c1-150; done
