set -x
mkdir -p gpurun_out/r6_call1
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_x3_gpu.py -m gpu -x -q -s -p no:cacheprovider > gpurun_out/r6_call1/tests_x3.log 2>&1
tail -15 gpurun_out/r6_call1/tests_x3.log
SAMAUDIO_HOSTILE_SIZE='large*' timeout 1200 python -m pytest tests/test_x3_gpu.py -m gpu -x -q -s -p no:cacheprovider -k hostile > gpurun_out/r6_call1/hostile_large.log 2>&1
tail -5 gpurun_out/r6_call1/hostile_large.log
timeout 900 python bench.py --precision fp16x3 --steps 3 --warmup 1 --no-other-configs --no-cpu-baseline > gpurun_out/r6_call1/bench_x3_b32.log 2>&1
tail -3 gpurun_out/r6_call1/bench_x3_b32.log | cut -c1-3000
timeout 600 python bench.py --precision fp16x3 --batch 4 --steps 3 --warmup 1 --no-other-configs --no-cpu-baseline > gpurun_out/r6_call1/bench_x3_b4.log 2>&1
tail -1 gpurun_out/r6_call1/bench_x3_b4.log | cut -c1-1500
timeout 600 python bench.py --precision fp16x3 --size 'small*' --batch 8 --steps 3 --warmup 1 --no-other-configs --no-cpu-baseline > gpurun_out/r6_call1/bench_x3_small.log 2>&1
tail -1 gpurun_out/r6_call1/bench_x3_small.log | cut -c1-600
