#!/bin/bash
# GPU call 17 of round 3: row groups that encode their own clips, one codec phase apart (SAMAudio.stagger) - A/B against
# the common up-front encode (SAMAUDIO_NO_STAGGER=1) on one box, twice each, plus the tests of the multi-stream path.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3_call17
mkdir -p $O
( timeout 600 python -m pytest tests/test_path_gpu.py tests/test_zz_next_rows_gpu.py -m gpu -q ) > $O/gpu_tests_subset.log 2>&1; echo "tests exit=$?"; tail -3 $O/gpu_tests_subset.log
Q="--no-cpu-baseline --no-parity-mode --no-roofline --steps 8 --warmup 2"
for i in 1 2; do
  ( timeout 300 python bench.py $Q ) > $O/bench_stagger_$i.log 2>&1
  ( SAMAUDIO_NO_STAGGER=1 timeout 300 python bench.py $Q ) > $O/bench_upfront_$i.log 2>&1
done
for f in bench_stagger_1 bench_upfront_1 bench_stagger_2 bench_upfront_2; do echo $f; grep -o '"value": [0-9.]*' $O/$f.log | head -1; done
( timeout 300 python bench.py $Q --batch 4 ) > $O/bench_b4_stagger.log 2>&1
( SAMAUDIO_NO_STAGGER=1 timeout 300 python bench.py $Q --batch 4 ) > $O/bench_b4_upfront.log 2>&1
for f in bench_b4_stagger bench_b4_upfront; do echo $f; grep -o '"value": [0-9.]*' $O/$f.log | head -1; done
