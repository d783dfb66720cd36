#!/bin/bash
# Round-4 second closing set (mixed-precision headline, persistent kernels): the bench line as the driver runs it (mixed
# headline, pure bf16 side by side, other_configs), quick lines of the pure fp16 / bf16 modes, rocprofv3 kernel stats of the serialised command, PMC passes restricted to the
# dominant kernel's symbol (HBM traffic: FETCH_SIZE / WRITE_SIZE in separate runs; SQ counters), whole GPU suite.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4_final2; mkdir -p $O
export OMP_NUM_THREADS=16
( time timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_driver_like.log 2> $O/bench_driver_like.err
grep -o '"value": [0-9.]*' $O/bench_driver_like.log | head -3
Q="--no-cpu-baseline --no-parity-mode --no-other-configs"
for prec in fp16 bf16; do timeout 400 python bench.py $Q --no-roofline --steps 6 --warmup 2 --precision $prec > $O/bench_$prec.log 2>&1; echo "$prec: $(grep -o '"value": [0-9.]*' $O/bench_$prec.log | tail -1)"; done
( timeout 500 rocprofv3 --kernel-trace --stats -d $O/trace -o t -- python bench.py $Q --no-roofline --steps 2 --warmup 1 --serial-groups ) > $O/trace.log 2>&1
db=$(find $O/trace -name '*_results.db' | head -1); [ -n "$db" ] && python tools/rocpd_stats.py $db > $O/kernel_stats_serial.md 2>/dev/null; rm -rf $O/trace
P="--no-cpu-baseline --no-parity-mode --no-other-configs --no-roofline --steps 1 --warmup 0 --serial-groups"
for c in FETCH_SIZE WRITE_SIZE; do
  ( timeout 500 rocprofv3 --kernel-trace --pmc $c --kernel-include-regex 'gemm8_kernel' --output-format csv -d $O/a/pmc_$c -o p -- python bench.py $P ) > $O/pmc_$c.log 2>&1; echo "pmc $c exit=$?"
done
python tools/pmc_traffic.py $O/a > $O/traffic_gemm8.json 2>$O/traffic.err; rm -rf $O/a
( PROBE_ROWS=4000 timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $O/sq -o p -- python tools/gemm_probe.py 22:qkv 22:wo 22:c_wq 22:w13 22:w2 ) > $O/pmc_sq.log 2>&1
f=$(find $O/sq -name '*counter_collection.csv' | head -1); [ -n "$f" ] && cp $f $O/pmc_sq_counters.csv; rm -rf $O/sq
timeout 600 python tools/stress_determinism.py --reps 40 --config default --clips 4 --frames 250 --what forward --precision mixed > $O/stress_default_forward.log 2>&1; tail -1 $O/stress_default_forward.log
timeout 300 python tools/diag_rerun.py --runs 6 > $O/rerun_checksums.log 2>&1; tail -3 $O/rerun_checksums.log
( timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider ) > $O/gpu_tests.log 2>&1; echo "gpu suite exit=$?"; tail -2 $O/gpu_tests.log
