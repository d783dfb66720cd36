#!/bin/bash
# GPU call 22 of round 3: gemm8w - the 256x256 tile on four waves of 128x128 with the accumulators pinned by inline-assembly
# MFMAs.  Its bitwise tests against gemm8 / gemm8s on hardware, then the DiT shapes beside gemm8 and hipBLASLt.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3_call22
mkdir -p $O
( timeout 300 python -m pytest tests/test_gemm2_gpu.py -m gpu -q -k "8phase_family or pipelined_form or 36" ) > $O/tests_gemm8w.log 2>&1; echo "tests exit=$?"; tail -3 $O/tests_gemm8w.log
( timeout 400 python tools/gemm_bench.py --iters 20 --clips 32 16 ) > $O/gemm_bench.log 2>&1; echo "gemm_bench exit=$?"
cut -c1-400 $O/gemm_bench.log | tail -20
