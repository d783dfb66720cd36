#!/bin/bash
# GPU call 4 of round 3: the whole -m gpu suite on the pruned library (8-phase loop with the plain-GEMM specialisation, new
# fold kernel, text-tower switch), the full default bench line (roofline from the timed launches, parity mode side by side),
# and the rocprofv3 --kernel-trace --stats summaries of the timed command and of its serialised form.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3_call4
mkdir -p $O
export OMP_NUM_THREADS=16
( time timeout 1500 python -m pytest tests -m gpu -q -x ) > $O/gpu_tests.log 2>&1
( time timeout 900 python bench.py ) > $O/bench.log 2>&1
Q="--no-cpu-baseline --no-parity-mode --no-roofline --steps 2 --warmup 1"
( timeout 400 rocprofv3 --kernel-trace --stats -d $O/trace_2streams -o t -- python bench.py $Q ) > $O/trace_2streams.log 2>&1
( timeout 400 rocprofv3 --kernel-trace --stats -d $O/trace_serial -o t -- python bench.py $Q --serial-groups ) > $O/trace_serial.log 2>&1
for d in trace_2streams trace_serial; do
  db=$(find $O/$d -name '*_results.db' | head -1)
  [ -n "$db" ] && python tools/rocpd_stats.py $db > $O/kernel_stats_$d.md 2>/dev/null
  rm -rf $O/$d
done
tail -3 $O/gpu_tests.log; grep -o '"value": [0-9.]*' $O/bench.log | head -2; head -12 $O/kernel_stats_trace_serial.md
