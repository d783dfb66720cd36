#!/bin/bash
# Round 2, GPU call 38: the N-rank path on the final code in the only form a 1-GPU box allows - two ranks time-sharing cuda:0
# (process group on gloo): self-launch, bucketed weight broadcast, weak + strong sharding, barrier, max over ranks, JSON line.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r2_call38
mkdir -p $OUT
(timeout 400 python bench.py --gpus 2 --share-gpu --steps 2 --warmup 1 --no-cpu-baseline --no-roofline) > $OUT/bench_2ranks_shared_gpu.log 2>&1; echo exit=$?; tail -1 $OUT/bench_2ranks_shared_gpu.log | cut -c1-400
