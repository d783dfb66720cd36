#!/bin/bash
# Round 2, GPU call 15: the vision tower's GEMM shapes per kernel (K = 1024 is only 16 K-tiles per tile).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r2_call15
mkdir -p $OUT
(timeout 300 python tools/gemm_bench.py --vit) > $OUT/gemm_vit.log 2>&1; tail -4 $OUT/gemm_vit.log | cut -c1-330
