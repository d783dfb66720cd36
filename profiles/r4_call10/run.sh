#!/bin/bash
# round 4, GPU call 10: is ONE q|k|v-epilogue launch deterministic?  (call 9: with the fused epilogue the per-stage checksums
# of consecutive solves first differ at Q; with the separate pass nothing differs)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r4_call10; mkdir -p $O
timeout 300 python tools/diag_qkv.py > $O/diag_qkv.log 2>&1; grep -v amdgpu $O/diag_qkv.log
