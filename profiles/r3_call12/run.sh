#!/bin/bash
# GPU call 12 of round 3: (a) why tests/test_path_gpu.py::test_concurrent_streams_are_bitwise_equal_to_one_stream failed in
# call 11 (tools/diag_streams.py: sharding vs concurrency vs kernel form); (b) the residual-unit kernels with the
# direct-to-LDS loads issued as inline assembly (the compiler no longer parks s_waitcnt vmcnt(0) in front of every k-step of
# the ring kernel) and the reworked weight-stationary kernel: bitwise tests, units in isolation, quick bench A/B
# (flag 19 = 2: ring kernel only); (c) the rest of the GPU suite from the failing test on.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3_call12
mkdir -p $O
( timeout 300 python tools/diag_streams.py ) > $O/diag_streams.log 2>&1; echo "diag exit=$?"; grep "rep" $O/diag_streams.log | head -20
( timeout 300 python -m pytest tests/test_gemm2_gpu.py tests/test_path_gpu.py -m gpu -q -k "fused_residual" ) > $O/tests_resunit.log 2>&1; echo "resunit tests exit=$?"; tail -2 $O/tests_resunit.log
( timeout 300 python tools/op_bench.py ) > $O/op_bench.log 2>&1
grep "residual unit" $O/op_bench.log
Q="--no-cpu-baseline --no-parity-mode --steps 6 --warmup 2"
( timeout 300 python bench.py $Q ) > $O/bench_ws.log 2>&1
( SAMAUDIO_DEBUG_FLAGS=19=2 timeout 300 python bench.py $Q ) > $O/bench_ring.log 2>&1
for f in bench_ws bench_ring; do echo $f; grep -o '"value": [0-9.]*' $O/$f.log | head -1; done
( timeout 1500 python -m pytest tests -m gpu -q --deselect tests/test_path_gpu.py::test_concurrent_streams_are_bitwise_equal_to_one_stream ) > $O/gpu_tests.log 2>&1; echo "gpu suite (without the streams test) exit=$?"
tail -5 $O/gpu_tests.log
