#!/bin/bash
# GPU call 5 of round 3: residual operands of the 32x32x16-family epilogue requested ahead of the loop (one memory round
# trip per 32-row slice instead of one per iteration) - A/B against the previous library on the same box: DAC residual
# units in isolation (tools/op_bench.py) and the quick bench line; plus which hipBLASLt kernels win the yardstick shapes.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3_call5
mkdir -p $O
( timeout 400 python tools/op_bench.py ) > $O/op_bench_new.log 2>&1
( SAMAUDIO_LIB_AB=$PWD/sam_audio_amd/libsamaudio_hip_prev.so timeout 400 python tools/op_bench.py ) > $O/op_bench_prev.log 2>&1
Q="--no-cpu-baseline --no-parity-mode --steps 4 --warmup 1"
( timeout 300 python bench.py $Q ) > $O/bench_new.log 2>&1
( SAMAUDIO_LIB_AB=$PWD/sam_audio_amd/libsamaudio_hip_prev.so timeout 300 python bench.py $Q ) > $O/bench_prev.log 2>&1
( timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_blas -o t -- python tools/yardstick.py gemm --out $O/yardstick.json ) > $O/yardstick_gemm.log 2>&1
db=$(find $O/trace_blas -name '*_results.db' | head -1); [ -n "$db" ] && python tools/rocpd_stats.py $db > $O/kernel_stats_hipblaslt.md 2>/dev/null; rm -rf $O/trace_blas
grep "residual unit" $O/op_bench_new.log $O/op_bench_prev.log
for f in bench_new bench_prev; do echo $f; grep -o '"value": [0-9.]*' $O/$f.log | head -1; done
