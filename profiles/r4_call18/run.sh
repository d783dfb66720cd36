#!/bin/bash
# Round-4 call 18: the two-stream difference under the per-stage checksum trace, as many repetitions as five minutes hold (it showed
# once in 300 repetitions in the third closing set: the last clip of the FIRST row group this time) - which stage differs first?
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4_call18; mkdir -p $O
export OMP_NUM_THREADS=8
SAMAUDIO_POISON=1 timeout 290 python tools/diag_hash.py --reps 4000 --keep $O/hash > $O/diag_hash.log 2>&1; echo "exit=$?"; tail -5 $O/diag_hash.log | cut -c1-300
ls $O/hash 2>/dev/null | head
