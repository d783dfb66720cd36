#!/bin/bash
# GPU call 11 of round 3: the weight-stationary persistent residual-unit kernel (gemm2.hip resws_kernel).  Its bitwise
# tests on hardware, the units in isolation (tools/op_bench.py: two launches | ring kernel | weight-stationary), the quick
# bench line with and without it (debug flag 19 = 2: ring kernel), then the whole GPU suite on this code.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3_call11
mkdir -p $O
( timeout 300 python -m pytest tests/test_gemm2_gpu.py tests/test_path_gpu.py -m gpu -q -k "fused_residual" ) > $O/tests_resunit.log 2>&1; echo "resunit tests exit=$?"
( timeout 300 python tools/op_bench.py ) > $O/op_bench.log 2>&1
grep "residual unit" $O/op_bench.log
Q="--no-cpu-baseline --no-parity-mode --steps 6 --warmup 2"
( timeout 300 python bench.py $Q ) > $O/bench_ws.log 2>&1
( SAMAUDIO_DEBUG_FLAGS=19=2 timeout 300 python bench.py $Q ) > $O/bench_ring.log 2>&1
for f in bench_ws bench_ring; do echo $f; grep -o '"value": [0-9.]*' $O/$f.log | head -1; done
( timeout 1500 python -m pytest tests -m gpu -q -x ) > $O/gpu_tests.log 2>&1; echo "gpu suite exit=$?"
tail -3 $O/gpu_tests.log
