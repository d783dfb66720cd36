#!/bin/bash
# round 4, GPU call 6: the whole -m gpu suite on the asm-DMA 8-phase family (gemm8 / gemm8s), including the new parity tests
# at the other benchmarked configurations (tests/test_configs_gpu.py); then the two-stream test hunted with per-stage
# checksums (SAMAUDIO_TRACE_HASH=1, the sharding test of the same file before it, as in the sessions where it failed);
# the few-row GEMM shapes; bench lines (headline precision fp16 with bf16 side by side, other_configs).
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r4_call6; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=15 > $O/tests_all.log 2>&1; tail -3 $O/tests_all.log
for i in $(seq 1 12); do
  SAMAUDIO_TRACE_HASH=1 timeout 300 python -m pytest tests/test_path_gpu.py -m gpu -q -p no:cacheprovider -s -k "full_width or concurrent_streams" > $O/hunt_$i.log 2>&1
  if grep -q "failed" $O/hunt_$i.log; then echo "hunt $i: FAILED (log kept)"; break; else rm -f $O/hunt_$i.log; fi
done
echo "hunt done"
timeout 600 python tools/gemm_bench.py --clips 4 16 --iters 20 > $O/gemm_bench.log 2>&1
timeout 1700 python bench.py > $O/bench_default.log 2> $O/bench_default.err; tail -c 1500 $O/bench_default.log
