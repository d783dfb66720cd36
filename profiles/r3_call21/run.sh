#!/bin/bash
# GPU call 21 of round 3: cross_attn_fold's batch split - how many workgroups to aim for (the split re-reads Wo once per
# batch slice): 132 (no split), 264, 384, 512 (shipped), per launch at 32 / 16 / 4 clips; first run = one trip at a time, second run
# (this script as committed) = V fragments requested one trip ahead.  Experiment build (env switch).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3_call21
mkdir -p $O
for t in 100 264 384 512; do ( SAMAUDIO_FOLD_TARGET=$t timeout 120 python tools/fold_bench.py ) > $O/fold_target_$t.log 2>&1; echo "target $t"; grep -i "us\b\|us " $O/fold_target_$t.log | head -4; done
