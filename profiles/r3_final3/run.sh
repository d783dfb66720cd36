#!/bin/bash
# Round-3 third closing set (GPU call 25), on the final commit (after the fold split policy and the attention-kernel load
# changes): the bench line as the driver runs it, rocprofv3 kernel stats of the same (serialised) command, the 4-clip and
# small* lines, the whole GPU suite.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3_final3
mkdir -p $O
export OMP_NUM_THREADS=16
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_driver_like.log 2>&1
Q="--no-cpu-baseline --no-parity-mode"
( timeout 400 rocprofv3 --kernel-trace --stats -d $O/trace -o t -- python bench.py $Q --no-roofline --steps 2 --warmup 1 --serial-groups ) > $O/trace.log 2>&1
db=$(find $O/trace -name '*_results.db' | head -1); [ -n "$db" ] && python tools/rocpd_stats.py $db > $O/kernel_stats_serial.md 2>/dev/null; rm -rf $O/trace
( timeout 300 python bench.py $Q --batch 4 --steps 6 --warmup 2 ) > $O/bench_batch4.log 2>&1
( timeout 300 python bench.py $Q --size 'small*' --batch 8 --steps 6 --warmup 2 ) > $O/bench_config1_small_b8.log 2>&1
for f in bench_driver_like bench_batch4 bench_config1_small_b8; do echo $f; grep -o '"value": [0-9.]*' $O/$f.log | head -2; done
( timeout 1500 python -m pytest tests -m gpu -q ) > $O/gpu_tests.log 2>&1; echo "gpu suite exit=$?"
tail -3 $O/gpu_tests.log
