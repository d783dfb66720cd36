#!/bin/bash
# Re-run of what the driver runs at round end, on another box and on the round's last build (comment-only changes since r6_final6):
# smoke(), the GPU suite (default markers: no slow tests), a short bench line.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${R6_OUT:-r6_final7}; mkdir -p $O
export OMP_NUM_THREADS=16
( time timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ) > $O/smoke.log 2>&1; grep "smoke" $O/smoke.log | cut -c1-120
( timeout 2400 python -m pytest tests -x -q -m gpu -p no:cacheprovider ) > $O/gpu_tests.log 2>&1; echo "gpu suite exit=$?"; tail -2 $O/gpu_tests.log
( time timeout 900 python bench.py --no-other-configs ) > $O/bench_default.log 2> $O/bench_default.err; echo "bench exit=$?"; grep -o '"value": [0-9.]*' $O/bench_default.log | head -3; tail -3 $O/bench_default.err | grep real
