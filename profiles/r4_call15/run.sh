#!/bin/bash
# Round-4 call 15: is the 4-clip configuration waiting for the host?  rocprofv3 kernel trace of `bench.py --batch 4` (one stream):
# kernel time against the idle gaps between dispatches, with gemm8s' pipelined form without (flag 27 = 1) and with requesting
# waves (2) - if the launches get shorter and the step does not, the GPU is waiting for launches.  The same for small* 8 clips.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4_call15; mkdir -p $O
export OMP_NUM_THREADS=16
Q="--no-cpu-baseline --no-parity-mode --no-other-configs --no-roofline --steps 3 --warmup 1"
for r in 1 2; do
  ( SAMAUDIO_DEBUG_FLAGS=27=$r timeout 300 rocprofv3 --kernel-trace --stats -d $O/t$r -o t -- python bench.py $Q --batch 4 ) > $O/trace_b4_roles$r.log 2>&1
  db=$(find $O/t$r -name '*_results.db' | head -1); [ -n "$db" ] && python tools/rocpd_stats.py $db > $O/kernel_stats_b4_roles$r.md 2>/dev/null; rm -rf $O/t$r
  echo "flag 27=$r: $(grep -o '"value": [0-9.]*' $O/trace_b4_roles$r.log | tail -1) | $(tail -1 $O/kernel_stats_b4_roles$r.md)"
done
( timeout 300 rocprofv3 --kernel-trace --stats -d $O/ts -o t -- python bench.py $Q --size 'small*' --batch 8 ) > $O/trace_small.log 2>&1
db=$(find $O/ts -name '*_results.db' | head -1); [ -n "$db" ] && python tools/rocpd_stats.py $db > $O/kernel_stats_small.md 2>/dev/null; rm -rf $O/ts
echo "small*: $(tail -1 $O/kernel_stats_small.md)"
