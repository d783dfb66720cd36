#!/bin/bash
# Round 5, GPU call 6: the 16-bit modes on trained-like ("hostile") weights with the sentinel on - small* (the suite's test) and large*
# (the benchmarked dims) -, the T5-in-step cost after the blocking read left the prompt path, the new path tests.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5_call6; mkdir -p $O
export OMP_NUM_THREADS=16
( timeout 900 python -m pytest tests/test_hostile_gpu.py -m gpu -q -s -p no:cacheprovider ) > $O/hostile_small.log 2>&1; echo "hostile small* exit=$?"; grep "^hostile\|passed\|failed" $O/hostile_small.log | cut -c1-400
( SAMAUDIO_HOSTILE_SIZE='large*' timeout 1200 python -m pytest tests/test_hostile_gpu.py -m gpu -q -s -p no:cacheprovider ) > $O/hostile_large.log 2>&1; echo "hostile large* exit=$?"; grep "^hostile\|passed\|failed" $O/hostile_large.log | cut -c1-400
( timeout 600 python -m pytest tests/test_path_gpu.py tests/test_t5_gpu.py -m gpu -q -x -p no:cacheprovider -k "f32_class_changes or weight_layout or text_encoder_wrapper or optional" ) > $O/tests.log 2>&1; echo "tests exit=$?"; tail -2 $O/tests.log
Q="--no-cpu-baseline --no-parity-mode --no-other-configs --no-verify"
for v in "b32_t5 --steps 6 --warmup 2" "b32_not5 --steps 6 --warmup 2 --no-t5" "b4_t5 --batch 4 --steps 8 --warmup 2" "b4_not5 --batch 4 --steps 8 --warmup 2 --no-t5" "b32_t5_again --steps 6 --warmup 2"; do
  set -- $v; name=$1; shift
  ( timeout 300 python bench.py $Q "$@" ) > $O/$name.log 2> $O/$name.err
  echo "$name $(grep -o '"value": [0-9.]*' $O/$name.log | head -1) $(grep -o '"ms_per_step": [0-9.]*' $O/$name.log | head -1)"
done
