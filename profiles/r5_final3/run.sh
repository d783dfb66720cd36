#!/bin/bash
# Round 5, last GPU call: smoke + the whole GPU suite on the final sources (the engine gained SAMAUDIO_OPT_ODE_GRAPH after r5_final2).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5_final3; mkdir -p $O
export OMP_NUM_THREADS=16
( timeout 300 python -c 'import __graft_entry__ as g; g.smoke()' ) > $O/smoke.log 2>&1; echo "smoke exit=$?"; tail -2 $O/smoke.log
( timeout 900 python -m pytest tests -m gpu -q ) > $O/gpu_tests.log 2>&1; echo "gpu suite exit=$?"; tail -4 $O/gpu_tests.log
