#!/bin/bash
# GPU call 13 of round 3: (a) the two-streams bitwise test that failed once in call 11 (and not in 15 solves of
# tools/diag_streams.py in call 12): six runs on its own, then the whole of tests/test_path_gpu.py in file order; (b) the
# quick bench line with all residual units fused (ring kernel for 64 / 128 / 192 channels, weight-stationary for 96) and,
# for comparison, as two launches each (debug flag 16); (c) the whole GPU suite.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3_call13
mkdir -p $O
for i in 1 2 3 4 5 6; do ( timeout 200 python -m pytest tests/test_path_gpu.py -m gpu -q -k "concurrent_streams" ) > $O/streams_$i.log 2>&1; echo "streams run $i exit=$?"; done
( timeout 600 python -m pytest tests/test_path_gpu.py -m gpu -q ) > $O/test_path.log 2>&1; echo "test_path_gpu exit=$?"; tail -3 $O/test_path.log
Q="--no-cpu-baseline --no-parity-mode --steps 6 --warmup 2"
( timeout 300 python bench.py $Q ) > $O/bench_fused.log 2>&1
( SAMAUDIO_DEBUG_FLAGS=16=1 timeout 300 python bench.py $Q ) > $O/bench_two_launches.log 2>&1
for f in bench_fused bench_two_launches; do echo $f; grep -o '"value": [0-9.]*' $O/$f.log | head -1; done
( timeout 1500 python -m pytest tests -m gpu -q ) > $O/gpu_tests.log 2>&1; echo "gpu suite exit=$?"
tail -5 $O/gpu_tests.log
