#!/bin/bash
# Round 2, GPU call 31: ModernBERT text tower of the Judge / PE-A-Frame on the HIP library - parity vs transformers'
# ModernBertModel (small configurations, ModernBERT-base dims in fp32 / bf16 / fp16), the Judge / span-predictor / checkpoint
# tests that now run through it, the whole -m gpu suite, configs[3] bench line.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r2_call31
mkdir -p $OUT
(timeout 600 python -m pytest tests/test_mbert_gpu.py -m gpu -q -s) > $OUT/gpu_tests_mbert.log 2>&1; tail -2 $OUT/gpu_tests_mbert.log; grep -E "^mbert|\.mbert" $OUT/gpu_tests_mbert.log | cut -c1-150
(timeout 900 python -m pytest tests -m gpu -q) > $OUT/gpu_tests.log 2>&1; tail -2 $OUT/gpu_tests.log
b() { name=$1; shift; (timeout 400 python bench.py --no-cpu-baseline "$@") > $OUT/bench_$name.log 2>&1; echo "$name $(tail -1 $OUT/bench_$name.log | cut -c52-100)"; }
b rerank_b8 --batch 8 --candidates 8 --predict-spans --steps 2
