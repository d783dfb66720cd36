set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6_final5; mkdir -p $O
export OMP_NUM_THREADS=16 SAMAUDIO_SLOW_TESTS=1
( SAMAUDIO_SHAPES_SIZE='large*' timeout 2400 python -m pytest tests/test_zz_benchmarked_shapes_gpu.py -m gpu -q -s -p no:cacheprovider -k eight_candidates ) > $O/shapes_large.log 2>&1; grep "configs\[\|passed\|failed" $O/shapes_large.log | cut -c1-500
( timeout 900 python -m pytest tests/test_zz_benchmarked_shapes_gpu.py tests/test_x3_gpu.py -m gpu -q -s -p no:cacheprovider ) > $O/shapes_small_x3.log 2>&1; grep "configs\[\|passed\|failed" $O/shapes_small_x3.log | cut -c1-300
