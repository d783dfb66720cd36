#!/bin/bash
# Round 2, GPU call 17: the N-rank path of bench.py on hardware - 2 ranks time-sharing the one GPU of this box over gloo
# (functional check: weight broadcast, weak + strong sharding, barrier, max-over-ranks, JSON line), small dims to keep it short.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r2_call17
mkdir -p $OUT
(timeout 500 python bench.py --gpus 2 --share-gpu --size 'small*' --batch 8 --steps 3 --no-cpu-baseline) > $OUT/bench_2ranks_one_gpu_small.log 2>&1; tail -4 $OUT/bench_2ranks_one_gpu_small.log | cut -c1-400
(timeout 600 python bench.py --gpus 2 --share-gpu --batch 16 --steps 2 --no-cpu-baseline --no-roofline) > $OUT/bench_2ranks_one_gpu_large.log 2>&1; tail -3 $OUT/bench_2ranks_one_gpu_large.log | cut -c1-400
