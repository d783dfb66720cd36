#!/bin/bash
# round 4, GPU call 11: tail split of the 8-phase launches with two row groups (shipped: off with two groups; qkv at 4 000
# rows = 528 tiles = 2 rounds + 16 tiles costs a third tile time on the persistent kernel) - quick bench lines A/B with the
# roofline leg, twice each.
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r4_call11; mkdir -p $O
for v in 0 1 0 1; do SAMAUDIO_BENCH_TAIL_SPLIT=$v timeout 600 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-parity-mode --no-other-configs > $O/bench_ts${v}_$RANDOM.log 2>&1; f=$(ls -t $O/bench_ts${v}_*.log | head -1); echo "tail split $v: $(grep -o '"value": [0-9.]*' $f | head -1) $(grep -o '"frac": [0-9.]*' $f | head -1)"; done
