#!/bin/bash
# Round-4 call 20 (the round's last GPU seconds): qkv_prep on its own, rewritten input before every launch, second stream busy - does its
# Q output differ run to run outside the model, and what do the differing rows look like?
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4_call20; mkdir -p $O
timeout 80 python tools/stress_qkv_prep.py --iters 250000 > $O/stress_qkv_prep.log 2>&1; echo "exit=$?"; tail -9 $O/stress_qkv_prep.log | cut -c1-400
