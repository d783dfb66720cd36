#!/bin/bash
# Round 2, GPU call 34: smoke() and the default bench line (CPU baseline + parity check) on the round's final code.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r2_call34
mkdir -p $OUT
(timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')") > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
(timeout 500 python bench.py) > $OUT/bench.log 2>&1; tail -1 $OUT/bench.log | cut -c1-200
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r2_call34/bench.log") if l.startswith("{")][-1])
print(d["parity_check"]); print(d["cpu_baseline"]["value"]); r = d["roofline"]; print(r["achieved"], r["frac"], r["avg_launch_us"], r["single_group_tail_split"]["frac"], r["whole_step"]["achieved"])
PY
