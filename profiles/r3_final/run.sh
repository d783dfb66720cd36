#!/bin/bash
# Round-3 closing set (GPU call 6): the bench line as the driver runs it, the PMC traffic passes of the serialised command
# (FETCH_SIZE and WRITE_SIZE in separate runs, kernel-trace only), the strong-scaling share (4 clips per GPU) with its
# kernel stats, BASELINE configs[1] / [3] / [4], SQ counters of the 8-phase loop, and the GEMM sweep beside hipBLASLt.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3_final
mkdir -p $O
export OMP_NUM_THREADS=16
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_driver_like.log 2>&1
Q="--no-cpu-baseline --no-parity-mode"
for c in FETCH_SIZE WRITE_SIZE; do
  ( timeout 500 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -o p -- python bench.py $Q --no-roofline --steps 1 --warmup 0 --serial-groups ) > $O/pmc_$c.log 2>&1; echo "pmc $c exit=$?"
done
python tools/pmc_traffic.py $O > $O/r3_traffic.json 2>$O/pmc_traffic.err; rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
( timeout 300 python bench.py $Q --batch 4 --steps 6 --warmup 2 ) > $O/bench_batch4.log 2>&1
( timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_b4 -o t -- python bench.py $Q --no-roofline --batch 4 --steps 2 --warmup 1 ) > $O/trace_b4.log 2>&1
db=$(find $O/trace_b4 -name '*_results.db' | head -1); [ -n "$db" ] && python tools/rocpd_stats.py $db > $O/kernel_stats_batch4.md 2>/dev/null; rm -rf $O/trace_b4
( timeout 300 python bench.py $Q --size 'small*' --batch 8 --steps 6 --warmup 2 ) > $O/bench_config1_small_b8.log 2>&1
( timeout 600 python bench.py $Q --batch 8 --candidates 8 --predict-spans --steps 3 --warmup 1 ) > $O/bench_config3_rerank_b8.log 2>&1
( timeout 600 python bench.py $Q --visual --batch 4 --steps 4 --warmup 1 ) > $O/bench_config4_visual_b4.log 2>&1
( timeout 300 python bench.py $Q --precision fp16 --steps 6 --warmup 2 ) > $O/bench_fp16.log 2>&1
( timeout 300 python tools/gemm_bench.py --iters 20 ) > $O/gemm_bench.log 2>&1
( timeout 300 python tools/fold_bench.py ) > $O/fold_bench.log 2>&1
bash tools/pmc_gemm.sh 22:w13 22:c_wq 22:qkv 22:w2 > $O/pmc_gemm.log 2>&1
mkdir -p $O/pmc_gemm8 && for p in p1 p2; do f=$(find gpurun_out/pmc_gemm/$p -name '*counter_collection.csv' | head -1); [ -n "$f" ] && cp $f $O/pmc_gemm8/$p.csv; done
python tools/pmc_sq.py gpurun_out/pmc_gemm > $O/pmc_gemm8.md 2>/dev/null; rm -rf gpurun_out/pmc_gemm
for f in bench_driver_like bench_batch4 bench_config1_small_b8 bench_config3_rerank_b8 bench_config4_visual_b4 bench_fp16; do echo $f; grep -o '"value": [0-9.]*' $O/$f.log | head -2; done
head -c 600 $O/r3_traffic.json
