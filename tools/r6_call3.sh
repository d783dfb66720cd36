set -x
O=gpurun_out/r6_call3
mkdir -p $O
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_x3_gpu.py -m gpu -q -s -p no:cacheprovider > $O/tests_x3.log 2>&1
tail -25 $O/tests_x3.log
true
true
SAMAUDIO_HOSTILE_SIZE='large*' timeout 1200 python -m pytest tests/test_x3_gpu.py -m gpu -x -q -s -p no:cacheprovider -k hostile > $O/hostile_large.log 2>&1
tail -3 $O/hostile_large.log
timeout 900 python bench.py --precision fp16x3 --steps 3 --warmup 1 --no-other-configs --no-cpu-baseline > $O/bench_x3_b32.log 2>&1
tail -1 $O/bench_x3_b32.log | cut -c1-400
timeout 600 python bench.py --precision fp16x3 --batch 4 --steps 3 --warmup 1 --no-other-configs --no-cpu-baseline > $O/bench_x3_b4.log 2>&1
tail -1 $O/bench_x3_b4.log | cut -c1-400
timeout 600 python bench.py --precision fp16x3 --size 'small*' --batch 8 --steps 3 --warmup 1 --no-other-configs --no-cpu-baseline > $O/bench_x3_small.log 2>&1
tail -1 $O/bench_x3_small.log | cut -c1-400
