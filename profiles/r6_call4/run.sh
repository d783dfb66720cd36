set -x
O=gpurun_out/r6_call4
mkdir -p $O
cd $GRAFT_REPO_ROOT
(time python bench.py --gpus 1 --steps 20 --warmup 5) > $O/bench_driver_like.log 2> $O/bench_driver_like.err
tail -3 $O/bench_driver_like.err
tail -1 $O/bench_driver_like.log | cut -c1-1500
python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/gpu_tests.log 2>&1
tail -8 $O/gpu_tests.log
