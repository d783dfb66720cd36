#!/bin/bash
# Round-end measurement recipe (run on the GPU box): GPU tests, bench.py, rocprofv3 kernel trace of the same command,
# and the two PMC passes (FETCH_SIZE, WRITE_SIZE - separate runs, kernel-trace only) behind roofline.traffic.
# Outputs land in gpurun_out/final/; tools/rocpd_stats.py and tools/pmc_traffic.py turn them into profiles/*.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/final
mkdir -p $OUT
timeout 600 python -m pytest tests -m gpu -q > $OUT/gpu_tests.log 2>&1; tail -3 $OUT/gpu_tests.log
(timeout 400 python bench.py) > $OUT/bench.log 2>&1; tail -1 $OUT/bench.log | cut -c1-400
(timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline) > $OUT/trace.log 2>&1; echo trace exit=$?
for c in FETCH_SIZE WRITE_SIZE; do
  (timeout 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$c -o p -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline) > $OUT/pmc_$c.log 2>&1; echo pmc $c exit=$?
done
ls -la $OUT $OUT/* | head -40
