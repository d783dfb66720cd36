#!/bin/bash
# Round 2, GPU call 4: the policy adopted from call 3 (8-phase family for every N >= 2048, gemm8s 128x128 tile of the same
# family for launches with few rows, 2 streams by default), its bench line, the rocprofv3 kernel trace and the two PMC
# passes behind roofline.traffic (single stream, as the instrumented roofline step), small per-GPU batches.
# NB: debug flag 7 changed meaning after call 3 (now: PREVIOUS policy = 8-phase only for N >= 4096).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r2_call4
mkdir -p $OUT
(timeout 300 python -m pytest tests/test_gemm2_gpu.py tests/test_path_gpu.py tests/test_large_gpu.py -m gpu -q -x) > $OUT/gpu_tests_subset.log 2>&1; tail -2 $OUT/gpu_tests_subset.log
(timeout 200 python tools/gemm_bench.py --family) > $OUT/gemm_family_m8000.log 2>&1; tail -6 $OUT/gemm_family_m8000.log | cut -c1-330
(timeout 200 python tools/gemm_bench.py --family --batch 4) > $OUT/gemm_family_m1000.log 2>&1; tail -6 $OUT/gemm_family_m1000.log | cut -c1-330
(timeout 200 python tools/gemm_bench.py --family --batch 16) > $OUT/gemm_family_m4000.log 2>&1; tail -6 $OUT/gemm_family_m4000.log | cut -c1-330
(timeout 400 python bench.py) > $OUT/bench.log 2>&1; tail -1 $OUT/bench.log | cut -c1-300
b() { name=$1; shift; (timeout 300 python bench.py --no-cpu-baseline "$@") > $OUT/bench_$name.log 2>&1; tail -1 $OUT/bench_$name.log | cut -c1-160; }
b streams1 --streams 1
SAMAUDIO_DEBUG_FLAGS=7=1 b prev_policy
b batch4 --batch 4 --steps 5
SAMAUDIO_DEBUG_FLAGS=6=1 b batch4_mblind --batch 4 --steps 5 --no-roofline
b batch4_graph --batch 4 --steps 5 --graph 1 --no-roofline
b batch16 --batch 16 --steps 4 --no-roofline
(timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python bench.py --steps 1 --warmup 1 --streams 1 --no-cpu-baseline --no-roofline) > $OUT/trace.log 2>&1; echo trace exit=$?
python tools/rocpd_stats.py $(find $OUT/trace -name '*results.db' | head -1) > $OUT/kernel_stats_streams1.md 2>$OUT/kernel_stats.err; head -12 $OUT/kernel_stats_streams1.md
rm -rf $OUT/trace
(timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace2 -o t -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline) > $OUT/trace2.log 2>&1; echo trace2 exit=$?
python tools/rocpd_stats.py $(find $OUT/trace2 -name '*results.db' | head -1) > $OUT/kernel_stats_default_2streams.md 2>>$OUT/kernel_stats.err; head -6 $OUT/kernel_stats_default_2streams.md
rm -rf $OUT/trace2
for c in FETCH_SIZE WRITE_SIZE; do
  (timeout 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$c -o p -- python bench.py --steps 1 --warmup 0 --streams 1 --no-cpu-baseline --no-roofline) > $OUT/pmc_$c.log 2>&1; echo pmc $c exit=$?
done
python tools/pmc_traffic.py $OUT > $OUT/r2_traffic.json 2>$OUT/traffic.err; head -c 1200 $OUT/r2_traffic.json
rm -rf $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
ls -la $OUT
