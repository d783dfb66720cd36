#!/bin/bash
# Round 2, GPU call 7: tail split of the 8-phase launches (whole rounds on gemm8 + 128x128 quadrant tail on gemm8s), the whole
# -m gpu suite incl. the fp16-operand tests, bench lines with and without the split (1 and 2 streams), rocprofv3 + PMC passes.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r2_call7
mkdir -p $OUT
(timeout 900 python -m pytest tests -m gpu -q -s) > $OUT/gpu_tests.log 2>&1; tail -3 $OUT/gpu_tests.log; grep -E "fp16|FAILED" $OUT/gpu_tests.log | cut -c1-250 | head -30
(timeout 200 python tools/gemm_bench.py --family) > $OUT/gemm_family_m8000.log 2>&1; tail -6 $OUT/gemm_family_m8000.log | cut -c1-200
(timeout 200 python tools/gemm_bench.py --family --batch 16) > $OUT/gemm_family_m4000.log 2>&1; tail -6 $OUT/gemm_family_m4000.log | cut -c1-200
b() { name=$1; shift; (timeout 400 python bench.py --no-cpu-baseline "$@") > $OUT/bench_$name.log 2>&1; tail -1 $OUT/bench_$name.log | cut -c1-160; }
b split_s2
b split_s1 --streams 1
SAMAUDIO_DEBUG_FLAGS=10=1 b nosplit_s2 --no-roofline
SAMAUDIO_DEBUG_FLAGS=10=1 b nosplit_s1 --streams 1 --no-roofline
b split_s2_again --no-roofline
(timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python bench.py --steps 1 --warmup 1 --streams 1 --no-cpu-baseline --no-roofline) > $OUT/trace.log 2>&1; echo trace exit=$?
python tools/rocpd_stats.py $(find $OUT/trace -name '*results.db' | head -1) > $OUT/kernel_stats_streams1.md 2>$OUT/kernel_stats.err; head -12 $OUT/kernel_stats_streams1.md | cut -c1-160
rm -rf $OUT/trace
for c in FETCH_SIZE WRITE_SIZE; do
  (timeout 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$c -o p -- python bench.py --steps 1 --warmup 0 --streams 1 --no-cpu-baseline --no-roofline) > $OUT/pmc_$c.log 2>&1; echo pmc $c exit=$?
done
python tools/pmc_traffic.py $OUT > $OUT/r2_traffic.json 2>$OUT/traffic.err
rm -rf $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
ls $OUT
