#!/bin/bash
# Round 2, GPU call 30: HBM traffic of the final code's kernels for the PRIMARY roofline entry - the two PMC passes
# (FETCH_SIZE, WRITE_SIZE; separate runs, kernel trace only) of the single-stream command with the tail split off (flag 10),
# i.e. the launches the instrumented step of bench.py times.  -> profiles/r2_traffic.json
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r2_call30
mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE; do
  (SAMAUDIO_DEBUG_FLAGS=10=1 timeout 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$c -o p -- python bench.py --steps 1 --warmup 0 --streams 1 --no-cpu-baseline --no-roofline) > $OUT/pmc_$c.log 2>&1; echo pmc $c exit=$?
done
python tools/pmc_traffic.py $OUT > $OUT/r2_traffic.json 2>$OUT/traffic.err
rm -rf $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
python - <<'PY'
import json
d = json.load(open("gpurun_out/r2_call30/r2_traffic.json"))
for k, v in list(d["kernels"].items())[:8]: print(k[:60], v["launches"], round(v["traffic_bytes_per_launch"] / 1e6, 1), "MB")
PY
