#!/bin/bash
# Round-3 third closing set, part 2 (GPU call 26): BASELINE configs[3] and [4] and the fp16 line on the final commit.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3_final3
mkdir -p $O
Q="--no-cpu-baseline --no-parity-mode"
( timeout 600 python bench.py $Q --batch 8 --candidates 8 --predict-spans --steps 3 --warmup 1 ) > $O/bench_config3_rerank_b8.log 2>&1
( timeout 600 python bench.py $Q --visual --batch 4 --steps 4 --warmup 1 ) > $O/bench_config4_visual_b4.log 2>&1
( timeout 300 python bench.py $Q --precision fp16 --steps 6 --warmup 2 ) > $O/bench_fp16.log 2>&1
for f in bench_config3_rerank_b8 bench_config4_visual_b4 bench_fp16; do echo $f; grep -o '"value": [0-9.]*' $O/$f.log | head -1; done
