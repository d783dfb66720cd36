#!/bin/bash
# Round-6 closing set on the final code (the operand-sharing kernel gemm8x_kernel under the headline mode).  First the PMC passes
# restricted to the dominant kernel's symbol (HBM traffic of its K' = 3K launches: FETCH_SIZE / WRITE_SIZE in separate runs) - their
# table is what bench.py's roofline.traffic quotes, so it is refreshed BEFORE the bench line - then what the driver runs (smoke(), the
# default bench line with 20 steps), the full-size race checks (tools/x3_probe.py: repeatability + the 128 x 128 kernel's bits;
# tools/stress_determinism.py on a large* DiT evaluation of 16 clips), rocprofv3 kernel stats of the serialised command and of the
# 4-clip command, SQ counters, the 2-rank run sharing this GPU, the whole GPU suite, the benchmarked-shape tests.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${R6_OUT:-r6_final6}; mkdir -p $O
export OMP_NUM_THREADS=16
export SAMAUDIO_SLOW_TESTS=1   # tests/test_zz_benchmarked_shapes_gpu.py: the 8-candidate test too
Q="--no-cpu-baseline --no-parity-mode --no-other-configs --no-verify --no-hostile"
P="$Q --no-roofline --steps 1 --warmup 0 --serial-groups"
for c in FETCH_SIZE WRITE_SIZE; do
  ( timeout 600 rocprofv3 --kernel-trace --pmc $c --kernel-include-regex 'gemm8x?_kernel' --output-format csv -d $O/a/pmc_$c -o p -- python bench.py $P ) > $O/pmc_$c.log 2>&1; echo "pmc $c exit=$?"
done
python tools/pmc_traffic.py $O/a > $O/traffic_gemm8_x3.json 2>$O/traffic.err; rm -rf $O/a; head -20 $O/traffic_gemm8_x3.json
grep -q gemm8x_kernel $O/traffic_gemm8_x3.json && cp $O/traffic_gemm8_x3.json profiles/r6_traffic_x3.json
( time timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ) > $O/smoke.log 2>&1; grep "smoke" $O/smoke.log | cut -c1-120
( time timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_driver_like.log 2> $O/bench_driver_like.err; echo "bench exit=$?"
grep -o '"value": [0-9.]*' $O/bench_driver_like.log | head -3; tail -3 $O/bench_driver_like.err | grep real
python tools/x3_probe.py > $O/x3_probe.log 2>&1; cat $O/x3_probe.log | cut -c1-250
timeout 200 bash tools/clock_sample.sh $O/clock_sample.log; grep -i "sclk\|power\|sustained" $O/clock_sample.log | sort | uniq -c | sort -rn | head -12
( timeout 600 python tools/stress_determinism.py --reps 40 --config 'large*' --clips 16 --frames 250 --precision fp16x3 --what forward ) > $O/stress_forward_large.log 2>&1; tail -2 $O/stress_forward_large.log | cut -c1-200
( timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace -o t -- python bench.py $Q --no-roofline --steps 2 --warmup 1 --serial-groups ) > $O/trace.log 2>&1
db=$(find $O/trace -name '*_results.db' | head -1); [ -n "$db" ] && python tools/rocpd_stats.py $db > $O/kernel_stats_serial.md 2>$O/rocpd.err; rm -rf $O/trace
head -12 $O/kernel_stats_serial.md | cut -c1-160
( timeout 500 rocprofv3 --kernel-trace --stats -d $O/trace4 -o t -- python bench.py $Q --no-roofline --steps 2 --warmup 1 --batch 4 ) > $O/trace_b4.log 2>&1
db=$(find $O/trace4 -name '*_results.db' | head -1); [ -n "$db" ] && python tools/rocpd_stats.py $db > $O/kernel_stats_batch4.md 2>>$O/rocpd.err; rm -rf $O/trace4
( PROBE_ROWS=4000 PROBE_KMUL=3 PROBE_FLAGS=32768 timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $O/sq -o p -- python tools/gemm_probe.py 22:qkv 22:wo 22:c_wq 22:w13 22:w2 ) > $O/pmc_sq.log 2>&1
python tools/pmc_sq.py $O/sq > $O/pmc_sq_summary.md 2>$O/pmc_sq.err; f=$(find $O/sq -name '*counter_collection.csv' | head -1); [ -n "$f" ] && cp $f $O/pmc_sq_counters.csv; rm -rf $O/sq; cat $O/pmc_sq_summary.md | cut -c1-200
( timeout 900 python bench.py --gpus 2 --share-gpu --steps 3 --warmup 1 --no-cpu-baseline --no-parity-mode --no-other-configs --no-verify --no-roofline --no-hostile ) > $O/bench_2ranks_share_gpu.log 2> $O/bench_2ranks_share_gpu.err; echo "2-rank exit=$?"; grep -o '"value": [0-9.]*\|"n_gpus": [0-9]*' $O/bench_2ranks_share_gpu.log | head -3
( timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider ) > $O/gpu_tests.log 2>&1; echo "gpu suite exit=$?"; tail -2 $O/gpu_tests.log
if [ -z "$R6_SHORT" ]; then
( SAMAUDIO_SHAPES_SIZE='large*' timeout 2400 python -m pytest tests/test_zz_benchmarked_shapes_gpu.py -m gpu -q -s -p no:cacheprovider ) > $O/shapes_large.log 2>&1; grep "configs\[\|passed\|failed" $O/shapes_large.log | cut -c1-400
fi
