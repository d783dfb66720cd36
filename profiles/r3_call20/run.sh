#!/bin/bash
# GPU call 20 of round 3: what the driver runs at round end, on the final commit - smoke(), then bench.py with no flags.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3_call20
mkdir -p $O
( time timeout 600 python -c 'import __graft_entry__ as g; g.smoke()' ) > $O/smoke.log 2>&1; echo "smoke exit=$?"; tail -4 $O/smoke.log
( time timeout 900 python bench.py ) > $O/bench_default.log 2>&1; echo "bench exit=$?"; grep -o '"value": [0-9.]*' $O/bench_default.log | head -2; tail -4 $O/bench_default.log | grep real
