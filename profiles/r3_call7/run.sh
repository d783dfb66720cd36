#!/bin/bash
# GPU call 7 of round 3: HBM traffic of the dominant kernel.  The two PMC passes over the whole (serialised) bench command
# crashed inside rocprofv3's dispatch interception at the first gemm8s launch (profiles/r3_final/pmc_FETCH_SIZE.log), so:
# (a) the same passes restricted to the dominant kernel's symbol, (b) the passes over the five DiT GEMM shapes at the rows of
# the timed launches (tools/gemm_probe.py, PROBE_ROWS=4000), 3 launches each.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3_call7
mkdir -p $O
Q="--no-cpu-baseline --no-parity-mode --no-roofline --steps 1 --warmup 0 --serial-groups"
for c in FETCH_SIZE WRITE_SIZE; do
  ( timeout 500 rocprofv3 --kernel-trace --pmc $c --kernel-include-regex 'gemm8_kernel' --output-format csv -d $O/a/pmc_$c -o p -- python bench.py $Q ) > $O/a_pmc_$c.log 2>&1; echo "(a) pmc $c exit=$?"
done
python tools/pmc_traffic.py $O/a > $O/traffic_bench_gemm8_only.json 2>$O/a_traffic.err
for c in FETCH_SIZE WRITE_SIZE; do
  ( PROBE_ROWS=4000 timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/b/pmc_$c -o p -- python tools/gemm_probe.py 22:qkv 22:wo 22:c_wq 22:w13 22:w2 ) > $O/b_pmc_$c.log 2>&1; echo "(b) pmc $c exit=$?"
  f=$(find $O/b/pmc_$c -name '*counter_collection.csv' | head -1); [ -n "$f" ] && cp $f $O/probe_rows4000_$c.csv
done
python tools/pmc_traffic.py $O/b > $O/traffic_probe_rows4000.json 2>$O/b_traffic.err
rm -rf $O/a $O/b
head -c 900 $O/traffic_bench_gemm8_only.json; echo; head -c 900 $O/traffic_probe_rows4000.json
