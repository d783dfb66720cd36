#!/bin/bash
# Round-4 call 21 (the last GPU seconds of the round): qkv_prep on its own (call 20: its Q output differed in 30 % of 250 000 launches,
# K and V^T never) with the variants of its 16-bit rounding (debug flag 29), in the fp16 build, and without the second stream.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4_call21; mkdir -p $O
timeout 40 python tools/stress_qkv_prep.py --iters 40000 --variants 0 1 2 3 > $O/stress_bf16_variants.log 2>&1; echo "exit=$?"; grep -v "amdgpu.ids" $O/stress_bf16_variants.log | cut -c1-330
timeout 15 python tools/stress_qkv_prep.py --iters 40000 --variants 0 --operands fp16 > $O/stress_fp16.log 2>&1; grep -v "amdgpu.ids" $O/stress_fp16.log | cut -c1-330
timeout 15 python tools/stress_qkv_prep.py --iters 40000 --variants 0 --no-second-stream > $O/stress_bf16_one_stream.log 2>&1; grep -v "amdgpu.ids" $O/stress_bf16_one_stream.log | cut -c1-330
