#!/bin/bash
# Round 5, GPU call 4: the whole GPU suite on the new few-row defaults (K-tile-major weights in every 16-bit model, wave roles, prefetch
# chain; text towers HIP-only), and the 32-clip quick line for regression.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5_call4; mkdir -p $O
export OMP_NUM_THREADS=16
( timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider ) > $O/gpu_tests.log 2>&1; echo "gpu suite exit=$?"; tail -5 $O/gpu_tests.log
Q="--no-cpu-baseline --no-parity-mode --no-other-configs --no-verify"
( timeout 300 python bench.py $Q --steps 5 --warmup 2 ) > $O/b32_default.log 2> $O/b32_default.err; grep -o '"value": [0-9.]*' $O/b32_default.log | head -1
( SAMAUDIO_WEIGHT_LAYOUT=rows timeout 300 python bench.py $Q --steps 5 --warmup 2 ) > $O/b32_rows.log 2> $O/b32_rows.err; grep -o '"value": [0-9.]*' $O/b32_rows.log | head -1
