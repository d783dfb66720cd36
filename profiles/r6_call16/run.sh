set -x
O=gpurun_out/${R6_OUT:-r6_call16}
mkdir -p $O
cd $GRAFT_REPO_ROOT
Q="--no-cpu-baseline --no-parity-mode --no-other-configs --no-verify --no-roofline --no-hostile"
run() { n=$1; shift; env "$@" timeout 300 python bench.py $Q --steps 4 --warmup 1 $EXTRA > $O/bench_$n.log 2>&1; echo "$n: $(tail -1 $O/bench_$n.log | cut -c1-110)"; }
EXTRA="--streams 2" run s2 A=1
EXTRA="--streams 2" run s2_15_17 SAMAUDIO_GROUP_SIZES=15,17
EXTRA="--streams 2" run s2_17_15 SAMAUDIO_GROUP_SIZES=17,15
EXTRA="--streams 2" run s2_split SAMAUDIO_BENCH_TAIL_SPLIT=1
EXTRA="--streams 2" run s2_15_17_split SAMAUDIO_GROUP_SIZES=15,17 SAMAUDIO_BENCH_TAIL_SPLIT=1
EXTRA="--streams 3" run s3 A=1
EXTRA="--streams 3" run s3_split SAMAUDIO_BENCH_TAIL_SPLIT=1
EXTRA="--streams 4" run s4 A=1
EXTRA="--streams 1" run s1 A=1
EXTRA="--streams 2" run s2_again A=1
