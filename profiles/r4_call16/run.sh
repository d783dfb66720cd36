#!/bin/bash
# Round-4 call 16: what do warm weights buy?  Debug flag 28 reads every big GEMM's weights right before its launch (same stream); rocprofv3
# kernel durations of the GEMM kernels with and without it - at 4 clips (pipelined gemm8s without / with requesting waves) and at the
# headline configuration (32 clips, the two row groups one after the other).  Kernel durations only: the touch kernel's own time is extra.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4_call16; mkdir -p $O
export OMP_NUM_THREADS=16
Q="--no-cpu-baseline --no-parity-mode --no-other-configs --no-roofline --steps 2 --warmup 1"
run() {  # name, flags, bench args
  ( SAMAUDIO_DEBUG_FLAGS=$2 timeout 300 rocprofv3 --kernel-trace --stats -d $O/t_$1 -o t -- python bench.py $Q $3 ) > $O/trace_$1.log 2>&1
  db=$(find $O/t_$1 -name '*_results.db' | head -1); [ -n "$db" ] && python tools/rocpd_stats.py $db > $O/kernel_stats_$1.md 2>/dev/null; rm -rf $O/t_$1
  echo "== $1: $(grep -o '"value": [0-9.]*' $O/trace_$1.log | tail -1)"; grep "gemm8\|touch" $O/kernel_stats_$1.md | head -5 | cut -c1-130
}
run b4_cold 27=1 "--batch 4"
run b4_warm 27=1,28=1 "--batch 4"
run b4_roles_warm 27=2,28=1 "--batch 4"
run b32_cold 27=1 "--serial-groups"
run b32_warm 27=1,28=1 "--serial-groups"
