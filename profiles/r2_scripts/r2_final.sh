#!/bin/bash
# Round 2, closing measurement set (second run, after calls 18 - 21; run on the GPU box): the whole -m gpu suite, the default bench line (configs[2]) with CPU
# baseline and parity check, the other BASELINE configs, fp16, and - for the roofline's provenance - a rocprofv3 kernel trace
# plus the two PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, kernel-trace only) of the single-stream command.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r2_final2
mkdir -p $OUT
(timeout 900 python -m pytest tests -m gpu -q -s) > $OUT/gpu_tests.log 2>&1; tail -2 $OUT/gpu_tests.log
(timeout 400 python bench.py) > $OUT/bench.log 2>&1; tail -1 $OUT/bench.log | cut -c1-200
b() { name=$1; shift; (timeout 400 python bench.py --no-cpu-baseline "$@") > $OUT/bench_$name.log 2>&1; tail -1 $OUT/bench_$name.log | cut -c52-100; grep "vision tower:" $OUT/bench_$name.log; }
b streams1 --streams 1
b fp16 --precision fp16 --verify
b small_b8 --size 'small*' --batch 8 --steps 5
b rerank_b8 --batch 8 --candidates 8 --predict-spans --steps 2
b visual_b4 --visual --batch 4 --steps 3
b batch4 --batch 4 --steps 5
b t5 --t5 --no-roofline
(timeout 300 python tools/op_bench.py) > $OUT/op_bench.log 2>&1
(timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python bench.py --steps 1 --warmup 1 --streams 1 --no-cpu-baseline --no-roofline) > $OUT/trace.log 2>&1; echo trace exit=$?
python tools/rocpd_stats.py $(find $OUT/trace -name '*results.db' | head -1) > $OUT/kernel_stats_streams1.md 2>$OUT/kernel_stats.err; head -10 $OUT/kernel_stats_streams1.md | cut -c1-160
rm -rf $OUT/trace
for c in FETCH_SIZE WRITE_SIZE; do
  (timeout 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$c -o p -- python bench.py --steps 1 --warmup 0 --streams 1 --no-cpu-baseline --no-roofline) > $OUT/pmc_$c.log 2>&1; echo pmc $c exit=$?
done
python tools/pmc_traffic.py $OUT > $OUT/r2_traffic.json 2>$OUT/traffic.err
rm -rf $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
ls $OUT
