#!/bin/bash
# Round-5 second closing set, final commit, another box: smoke(), the driver's bench command, the whole GPU suite.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5_final2; mkdir -p $O
export OMP_NUM_THREADS=16
( time timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ) > $O/smoke.log 2>&1; grep "smoke" $O/smoke.log | cut -c1-120
( time timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_driver_like.log 2> $O/bench_driver_like.err; echo "bench exit=$?"
grep -o '"value": [0-9.]*' $O/bench_driver_like.log | head -3; tail -3 $O/bench_driver_like.err | grep real
( timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider ) > $O/gpu_tests.log 2>&1; echo "gpu suite exit=$?"; tail -2 $O/gpu_tests.log
