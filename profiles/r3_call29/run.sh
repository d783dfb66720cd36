#!/bin/bash
# GPU call 29 of round 3: HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) and SQ counters of the DiT's streaming kernels
# on tools/op_bench.py's launches (32 clips): self-attention, qkv_prep, RMSNorm+modulate, cross-attention, fold.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3_call29
mkdir -p $O
R='self_attn_bf16_kernel|qkv_prep_bf16_kernel|rmsnorm|cross_attn'
for c in FETCH_SIZE WRITE_SIZE; do
  ( timeout 300 rocprofv3 --kernel-trace --pmc $c --kernel-include-regex "$R" --output-format csv -d $O/pmc_$c -o p -- python tools/op_bench.py ) > $O/pmc_$c.log 2>&1; echo "pmc $c exit=$?"
done
python tools/pmc_traffic.py $O > $O/traffic_streaming.json 2>$O/traffic.err
( timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --kernel-include-regex "$R" --output-format csv -d $O/sq1 -o p -- python tools/op_bench.py ) > $O/sq1.log 2>&1; echo "sq1 exit=$?"
f=$(find $O/sq1 -name '*counter_collection.csv' | head -1); [ -n "$f" ] && cp $f $O/sq_counters.csv
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/sq1
python - <<'PY'
import json
d=json.load(open("gpurun_out/r3_call29/traffic_streaming.json"))
for k,v in d["kernels"].items(): print("%-70s launches %4d  %.1f MB per launch" % (k[:70], v["launches"], v["traffic_bytes_per_launch"]/1e6))
PY
