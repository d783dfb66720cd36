#!/bin/bash
# GPU call 24 of round 3: cross-attention kernels (probs, unfolded) with every load - mask bytes, q-norm weights, operands -
# requested up front and pinned there (was: mask bytes behind branches and q-norm weights one k-step at a time, each with its
# own vmcnt(0)).  Kernel tests, kernels in isolation, the quick bench line.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3_call24
mkdir -p $O
( timeout 400 python -m pytest tests/test_kernels_gpu.py tests/test_vit_gpu.py tests/test_zz_next_rows_gpu.py -m gpu -q ) > $O/gpu_tests_subset.log 2>&1; echo "tests exit=$?"; tail -2 $O/gpu_tests_subset.log
( timeout 300 python tools/op_bench.py ) > $O/op_bench.log 2>&1; grep -v "residual unit" $O/op_bench.log | tail -8
Q="--no-cpu-baseline --no-parity-mode --steps 6 --warmup 2"
( timeout 300 python bench.py $Q ) > $O/bench.log 2>&1
grep -o '"value": [0-9.]*' $O/bench.log | head -1
