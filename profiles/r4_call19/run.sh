#!/bin/bash
# Round-4 call 19: qkv_prep with the guarded DPP reduction (common.h row16_sum_guarded) - the run-to-run difference of call 18 (8 of
# 4 000 repetitions, all beginning in qkv_prep's Q output) under the same trace; then the GPU suite without the two tests whose CPU-oracle
# fixtures take 4 minutes, a quick bench line, and the full-solve parity test at large* dims.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4_call19; mkdir -p $O
export OMP_NUM_THREADS=16
SAMAUDIO_POISON=1 timeout 200 python tools/diag_hash.py --reps 4000 --keep $O/hash > $O/diag_hash.log 2>&1; echo "exit=$?"; tail -4 $O/diag_hash.log | cut -c1-250
( timeout 400 python -m pytest tests -m gpu -q -p no:cacheprovider -k "not full_solve" ) > $O/gpu_tests_most.log 2>&1; echo "suite (not full_solve) exit=$?"; tail -1 $O/gpu_tests_most.log
timeout 200 python bench.py --no-cpu-baseline --no-parity-mode --no-other-configs --no-roofline --steps 6 --warmup 2 > $O/bench_quick.log 2>&1; echo "quick: $(grep -o '"value": [0-9.]*' $O/bench_quick.log | tail -1)"
( timeout 300 python -m pytest tests/test_large_gpu.py -m gpu -q -p no:cacheprovider -k "full_solve" ) > $O/gpu_tests_large_full_solve.log 2>&1; echo "large full solve exit=$?"; tail -1 $O/gpu_tests_large_full_solve.log
