#!/bin/bash
# Round 5, GPU call 7: (a) which GEMM classes carry the 16-bit modes' error on the HOSTILE weights (error budget by operand-rounding
# emulation inside the fp32 engine, full solve at large*), (b) the head_dim-64 configuration on hardware (D = 1024, H = 16).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5_call7; mkdir -p $O
export OMP_NUM_THREADS=16
( timeout 600 python -m pytest tests/test_path_gpu.py tests/test_kernels_gpu.py -m gpu -q -x -s -p no:cacheprovider -k "head_dim_64 or qkv_prep or headnorm or cross_att or self_att" ) > $O/tests.log 2>&1; echo "tests exit=$?"; grep "head_dim 64\|passed\|failed" $O/tests.log | cut -c1-200
( timeout 1500 python tools/error_budget.py --hostile --out $O/error_budget_hostile.json ) > $O/error_budget_hostile.log 2>&1; echo "budget exit=$?"; cat $O/error_budget_hostile.log | cut -c1-200
