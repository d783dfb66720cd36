set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${R6_OUT:-r6_call27}
mkdir -p $O
export OMP_NUM_THREADS=16
SAMAUDIO_HOSTILE_SIZE='large*' timeout 2000 python -m pytest tests/test_x3_gpu.py tests/test_hostile_gpu.py -m gpu -q -s -p no:cacheprovider -k "hostile" > $O/hostile_large.log 2>&1
grep "hostile\|passed\|failed" $O/hostile_large.log | cut -c1-300
