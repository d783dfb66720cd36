#!/bin/bash
# round 4, GPU call 4: (a) what the 8-phase K loop spends its time on - ablation builds + s_memtime stamps of the real kernel
# (tools/gemm8_ablate.py on tools/abl/libsamaudio_hip_abl.so); (b) run-to-run determinism of the hot path's pieces on one
# stream, 300 repetitions each (tools/stress_determinism.py) - the two-stream bitwise test failed once in call 2 and once in
# round 3, both times inside a longer pytest process; (c) that test file 6 times in a row.
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r4_call4; mkdir -p $O
SAMAUDIO_LIB_AB=tools/abl/libsamaudio_hip_abl.so timeout 600 python tools/gemm8_ablate.py > $O/ablate.log 2>&1; tail -12 $O/ablate.log
timeout 900 python tools/stress_determinism.py --reps 300 > $O/stress_mini.log 2>&1; tail -4 $O/stress_mini.log
timeout 900 python tools/stress_determinism.py --reps 100 --config default --clips 3 --frames 40 --what encode decode forward > $O/stress_default.log 2>&1; tail -3 $O/stress_default.log
for i in 1 2 3 4 5 6; do timeout 600 python -m pytest tests/test_path_gpu.py -m gpu -q -p no:cacheprovider > $O/path_$i.log 2>&1; tail -1 $O/path_$i.log; done
