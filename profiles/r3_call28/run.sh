#!/bin/bash
# GPU call 28 of round 3: gemm8s' pipelined form on a 5-stage ring (160 KiB: all of the CU's LDS, four K-tiles in flight
# instead of three) - A/B against the 4-stage library on one box: few-row GEMM shapes, 4 clips per GPU, small*.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3_call28
mkdir -p $O
PREV=$PWD/sam_audio_amd/libsamaudio_hip_prev.so
( timeout 200 python -m pytest tests/test_gemm2_gpu.py -m gpu -q -k "8phase_family or pipelined_form or 27" ) > $O/tests.log 2>&1; echo "tests exit=$?"; tail -1 $O/tests.log
( timeout 300 python tools/gemm_bench.py --iters 20 --clips 4 --no-blas ) > $O/gemm_bench_5stage.log 2>&1
( SAMAUDIO_LIB_AB=$PREV timeout 300 python tools/gemm_bench.py --iters 20 --clips 4 --no-blas ) > $O/gemm_bench_4stage.log 2>&1
Q="--no-cpu-baseline --no-parity-mode --no-roofline --steps 8 --warmup 2"
for i in 1 2; do
  ( timeout 300 python bench.py $Q --batch 4 ) > $O/bench_b4_5stage_$i.log 2>&1
  ( SAMAUDIO_LIB_AB=$PREV timeout 300 python bench.py $Q --batch 4 ) > $O/bench_b4_4stage_$i.log 2>&1
done
( timeout 300 python bench.py $Q --size 'small*' --batch 8 ) > $O/bench_small_5stage.log 2>&1
( SAMAUDIO_LIB_AB=$PREV timeout 300 python bench.py $Q --size 'small*' --batch 8 ) > $O/bench_small_4stage.log 2>&1
for f in bench_b4_5stage_1 bench_b4_4stage_1 bench_b4_5stage_2 bench_b4_4stage_2 bench_small_5stage bench_small_4stage; do echo $f; grep -o '"value": [0-9.]*' $O/$f.log | head -1; done
grep "M=1000" $O/gemm_bench_5stage.log | cut -c1-260; echo; grep "M=1000" $O/gemm_bench_4stage.log | cut -c1-260
