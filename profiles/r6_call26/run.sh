set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${R6_OUT:-r6_call26}
mkdir -p $O
Q="--no-cpu-baseline --no-parity-mode --no-other-configs --no-hostile --no-verify --no-roofline"
v() { tail -1 $1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$1', d['value'], d['ms_per_step'], d['config'].get('streams_per_gpu'))"; }
timeout 400 python bench.py $Q --steps 4 --warmup 1 > $O/b32_default.log 2>&1; v $O/b32_default.log
SAMAUDIO_BENCH_TAIL_SPLIT=1 timeout 400 python bench.py $Q --steps 4 --warmup 1 > $O/b32_tailsplit.log 2>&1; v $O/b32_tailsplit.log
timeout 400 python bench.py $Q --steps 4 --warmup 1 --streams 3 > $O/b32_streams3.log 2>&1; v $O/b32_streams3.log
timeout 400 python bench.py $Q --steps 4 --warmup 1 --streams 1 > $O/b32_streams1.log 2>&1; v $O/b32_streams1.log
timeout 300 python bench.py $Q --steps 4 --warmup 1 --batch 4 > $O/b4_default.log 2>&1; v $O/b4_default.log
SAMAUDIO_DEBUG_FLAGS=30=100 timeout 300 python bench.py $Q --steps 4 --warmup 1 --batch 4 > $O/b4_gemm8_from100.log 2>&1; v $O/b4_gemm8_from100.log
SAMAUDIO_DEBUG_FLAGS=30=60 timeout 300 python bench.py $Q --steps 4 --warmup 1 --batch 4 > $O/b4_gemm8_from60.log 2>&1; v $O/b4_gemm8_from60.log
timeout 300 python bench.py $Q --steps 4 --warmup 1 --batch 8 > $O/b8_default.log 2>&1; v $O/b8_default.log
SAMAUDIO_DEBUG_FLAGS=30=100 timeout 300 python bench.py $Q --steps 4 --warmup 1 --batch 8 > $O/b8_gemm8_from100.log 2>&1; v $O/b8_gemm8_from100.log
timeout 300 python bench.py $Q --steps 4 --warmup 1 --batch 16 > $O/b16_default.log 2>&1; v $O/b16_default.log
