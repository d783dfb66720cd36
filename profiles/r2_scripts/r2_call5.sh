#!/bin/bash
# Round 2, GPU call 5: the PE-Core vision tower (rows a4 / f3) on hardware - the whole -m gpu suite (incl. tests/test_vit_gpu.py
# at PE-Core-L14-336 dims), BASELINE configs[4] (visual prompting, batch 4) with its rocprofv3 kernel trace, configs[3]
# (8 candidates + Judge + span predictor, batch 8) and configs[1] (small*, batch 8) re-measured with the new policy.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r2_call5
mkdir -p $OUT
(timeout 900 python -m pytest tests -m gpu -q -x -s) > $OUT/gpu_tests.log 2>&1; tail -3 $OUT/gpu_tests.log; grep -E "^vit |PE-Core" $OUT/gpu_tests.log | head -30
b() { name=$1; shift; (timeout 400 python bench.py --no-cpu-baseline "$@") > $OUT/bench_$name.log 2>&1; tail -1 $OUT/bench_$name.log | cut -c1-200; grep "vision tower:" $OUT/bench_$name.log; }
b visual_b4 --visual --batch 4 --steps 3
b rerank_b8 --batch 8 --candidates 8 --predict-spans --steps 2
b small_b8 --size 'small*' --batch 8 --steps 5
(timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python bench.py --visual --batch 4 --steps 1 --warmup 1 --no-cpu-baseline --no-roofline) > $OUT/trace.log 2>&1; echo trace exit=$?
python tools/rocpd_stats.py $(find $OUT/trace -name '*results.db' | head -1) > $OUT/kernel_stats_visual_b4.md 2>$OUT/kernel_stats.err; head -24 $OUT/kernel_stats_visual_b4.md | cut -c1-180
rm -rf $OUT/trace
ls -la $OUT
