#!/bin/bash
# Round 5, GPU call 8: the cross-attention fold of all layers in one launch per evaluation (default) against one launch per layer
# (debug flag 31 = 1), fp16 headline, at 32 / 4 clips and small* 8 clips; its bitwise test and the hostile bounds.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5_call8; mkdir -p $O
export OMP_NUM_THREADS=16
( timeout 600 python -m pytest tests/test_path_gpu.py tests/test_hostile_gpu.py -m gpu -q -x -p no:cacheprovider -k "fold_of_all or hostile or golden" ) > $O/tests.log 2>&1; echo "tests exit=$?"; tail -2 $O/tests.log
Q="--no-cpu-baseline --no-parity-mode --no-other-configs --no-verify"
run() { name=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  ( env "${envs[@]}" timeout 300 python bench.py $Q "$@" ) > $O/$name.log 2> $O/$name.err
  echo "$name $(grep -o '"value": [0-9.]*' $O/$name.log | head -1) $(grep -o '"ms_per_step": [0-9.]*' $O/$name.log | head -1)"; }
run b4_fold_all      X=1 -- --batch 4 --steps 8 --warmup 2
run b4_fold_layer    SAMAUDIO_DEBUG_FLAGS=31=1 -- --batch 4 --steps 8 --warmup 2
run b4_fold_all_2    X=1 -- --batch 4 --steps 8 --warmup 2
run b4_mixed         X=1 -- --batch 4 --steps 8 --warmup 2 --precision mixed
run s8_fold_all      X=1 -- --size 'small*' --batch 8 --steps 8 --warmup 2
run s8_fold_layer    SAMAUDIO_DEBUG_FLAGS=31=1 -- --size 'small*' --batch 8 --steps 8 --warmup 2
run b32_fold_all     X=1 -- --steps 6 --warmup 2
run b32_fold_layer   SAMAUDIO_DEBUG_FLAGS=31=1 -- --steps 6 --warmup 2
run b32_fold_all_2   X=1 -- --steps 6 --warmup 2
run b32_mixed        X=1 -- --steps 6 --warmup 2 --precision mixed
