#!/bin/bash
# round 4, GPU call 9: run-to-run difference seen in call 8 (test_batch_sharding...: "run-to-run determinism") with the fused
# q|k|v epilogue - per-stage checksums of consecutive runs, fused vs the separate pass (flag 27) vs one workgroup per tile (26).
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r4_call9; mkdir -p $O
timeout 300 python tools/diag_rerun.py --runs 4 > $O/rerun_fused.log 2>&1; cat $O/rerun_fused.log | tail -6
timeout 300 python tools/diag_rerun.py --runs 4 --flags 27=1 > $O/rerun_unfused.log 2>&1; cat $O/rerun_unfused.log | tail -6
timeout 300 python tools/diag_rerun.py --runs 4 --flags 26=1 > $O/rerun_nopersist.log 2>&1; cat $O/rerun_nopersist.log | tail -6
timeout 300 python tools/diag_rerun.py --runs 4 --flags 21=1 > $O/rerun_plainform.log 2>&1; cat $O/rerun_plainform.log | tail -6
