#!/bin/bash
# Round 2, GPU call 36: tile-raster group size (M-tiles per XCD group, default 8) of the 8-phase kernel on the DiT shapes,
# at 32 clips (M = 8000) and at 16 (M = 4000: one of two row groups).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r2_call36
mkdir -p $OUT
for bt in 32 16; do
  (timeout 300 python tools/gemm_bench.py --raster --batch $bt --iters 10) > $OUT/raster_b$bt.log 2>&1
  grep -E "M=[0-9]+ N=" $OUT/raster_b$bt.log | grep -v edge | sed -E 's/ +/ /g' | cut -c1-140
done
