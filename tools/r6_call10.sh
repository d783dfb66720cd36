set -x
O=gpurun_out/r6_call10
mkdir -p $O
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_x3_gpu.py tests/test_hostile_gpu.py -m gpu -q -s -p no:cacheprovider > $O/tests_x3_hostile.log 2>&1
grep "hostile\|cross-attention\|passed\|failed" $O/tests_x3_hostile.log | cut -c1-260
SAMAUDIO_HOSTILE_SIZE='large*' timeout 1500 python -m pytest tests/test_x3_gpu.py -m gpu -x -q -s -p no:cacheprovider -k hostile > $O/hostile_large.log 2>&1
grep "hostile\|passed\|failed" $O/hostile_large.log | cut -c1-300
Q="--no-cpu-baseline --no-parity-mode --no-other-configs --no-verify --no-hostile"
timeout 400 python bench.py $Q --steps 4 --warmup 1 > $O/bench_x3_b32.log 2>&1; tail -1 $O/bench_x3_b32.log | cut -c1-200
SAMAUDIO_NO_FOLD=1 timeout 400 python bench.py $Q --steps 4 --warmup 1 > $O/bench_x3_b32_nofold.log 2>&1; tail -1 $O/bench_x3_b32_nofold.log | cut -c1-200
for b in 4 16; do timeout 300 python bench.py $Q --no-roofline --steps 3 --warmup 1 --batch $b > $O/bench_x3_b$b.log 2>&1; tail -1 $O/bench_x3_b$b.log | cut -c1-160; done
