set -x
O=gpurun_out/r6_call5
mkdir -p $O
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_zz_benchmarked_shapes_gpu.py -m gpu -q -s -p no:cacheprovider > $O/shapes_small.log 2>&1
grep "configs\[\|passed\|failed" $O/shapes_small.log | cut -c1-400
python -m pytest tests/test_configs_gpu.py tests/test_zz_next_rows_gpu.py tests/test_vit_gpu.py -m gpu -q -s -p no:cacheprovider -k "judge or frame or vit or tower" > $O/tower_errors.log 2>&1
grep "max-abs err\|passed\|failed" $O/tower_errors.log | cut -c1-200 | tail -40
Q="--no-cpu-baseline --no-parity-mode --no-other-configs --no-verify --no-roofline --no-hostile"
timeout 300 python bench.py $Q --steps 4 --warmup 1 --streams 1 > $O/bench_x3_streams1.log 2>&1; tail -1 $O/bench_x3_streams1.log | cut -c1-200
timeout 300 python bench.py $Q --steps 4 --warmup 1 --streams 2 > $O/bench_x3_streams2.log 2>&1; tail -1 $O/bench_x3_streams2.log | cut -c1-200
SAMAUDIO_BENCH_TAIL_SPLIT=1 timeout 300 python bench.py $Q --steps 4 --warmup 1 --streams 2 > $O/bench_x3_streams2_split.log 2>&1; tail -1 $O/bench_x3_streams2_split.log | cut -c1-200
