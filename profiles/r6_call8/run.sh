set -x
O=gpurun_out/r6_call8
mkdir -p $O
cd $GRAFT_REPO_ROOT
Q="--no-cpu-baseline --no-parity-mode --no-other-configs --no-verify --no-hostile"
for gm in 0 4 2 16; do
  SAMAUDIO_DEBUG_FLAGS="35=$gm" SAMAUDIO_PROF_BY_CLASS=1 timeout 400 python bench.py $Q --steps 3 --warmup 1 > $O/bench_x3_gm$gm.log 2>&1; tail -1 $O/bench_x3_gm$gm.log | cut -c1-120
done
