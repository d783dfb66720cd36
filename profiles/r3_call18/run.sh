#!/bin/bash
# GPU call 18 of round 3: what bounds the DAC residual-unit kernels now - HBM traffic (FETCH_SIZE / WRITE_SIZE in separate
# passes) and SQ counters of the ring and the weight-stationary kernel on tools/op_bench.py's launches (8 waveforms).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3_call18
mkdir -p $O
R='res(unit|ws)_kernel'
for c in FETCH_SIZE WRITE_SIZE; do
  ( timeout 300 rocprofv3 --kernel-trace --pmc $c --kernel-include-regex "$R" --output-format csv -d $O/pmc_$c -o p -- python tools/op_bench.py ) > $O/pmc_$c.log 2>&1; echo "pmc $c exit=$?"
done
python tools/pmc_traffic.py $O > $O/traffic_resunit.json 2>$O/traffic.err
( timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --kernel-include-regex "$R" --output-format csv -d $O/sq1 -o p -- python tools/op_bench.py ) > $O/sq1.log 2>&1; echo "sq1 exit=$?"
f=$(find $O/sq1 -name '*counter_collection.csv' | head -1); [ -n "$f" ] && cp $f $O/sq_counters.csv
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/sq1
head -c 1500 $O/traffic_resunit.json
