set -x
O=gpurun_out/r6_call7
mkdir -p $O
cd $GRAFT_REPO_ROOT
Q="--no-cpu-baseline --no-parity-mode --no-other-configs --no-verify --no-hostile"
SAMAUDIO_PROF_BY_CLASS=1 timeout 400 python bench.py $Q --steps 2 --warmup 1 > $O/bench_x3_b32_byclass.log 2>&1; tail -1 $O/bench_x3_b32_byclass.log | cut -c1-200
SAMAUDIO_PROF_BY_CLASS=1 timeout 400 python bench.py $Q --steps 2 --warmup 1 --precision fp16 > $O/bench_fp16_b32_byclass.log 2>&1; tail -1 $O/bench_fp16_b32_byclass.log | cut -c1-200
