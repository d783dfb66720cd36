#!/bin/bash
# Round-4 third closing set, on the final commit: what the driver runs (smoke(), the default bench line with 20 steps, the whole GPU
# suite), the rocprofv3 kernel stats of the serialised command the roofline's per-launch durations must agree with, and the
# two-stream scenario 300 times.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4_final3; mkdir -p $O
export OMP_NUM_THREADS=16
( time timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ) > $O/smoke.log 2>&1; tail -1 $O/smoke.log | cut -c1-100
( time timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_driver_like.log 2> $O/bench_driver_like.err
grep -o '"value": [0-9.]*' $O/bench_driver_like.log | head -3
Q="--no-cpu-baseline --no-parity-mode --no-other-configs"
( timeout 500 rocprofv3 --kernel-trace --stats -d $O/trace -o t -- python bench.py $Q --no-roofline --steps 2 --warmup 1 --serial-groups ) > $O/trace.log 2>&1
db=$(find $O/trace -name '*_results.db' | head -1); [ -n "$db" ] && python tools/rocpd_stats.py $db > $O/kernel_stats_serial.md 2>/dev/null; rm -rf $O/trace
head -7 $O/kernel_stats_serial.md | tail -3 | cut -c1-120
SAMAUDIO_POISON=1 timeout 400 python tools/stress_two_streams.py --reps 300 > $O/stress_two_streams.log 2>&1; tail -1 $O/stress_two_streams.log
( timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider ) > $O/gpu_tests.log 2>&1; echo "gpu suite exit=$?"; tail -2 $O/gpu_tests.log
