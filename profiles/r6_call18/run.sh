set -x
O=gpurun_out/${R6_OUT:-r6_call18}
mkdir -p $O
cd $GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 python tools/vit_probe.py > $O/vit_probe.log 2>&1; tail -2 $O/vit_probe.log
( timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace -o t -- python tools/vit_probe.py ) > $O/trace.log 2>&1
db=$(find $O/trace -name '*_results.db' | head -1); [ -n "$db" ] && python tools/rocpd_stats.py $db > $O/kernel_stats_vit.md 2>$O/rocpd.err; rm -rf $O/trace
head -30 $O/kernel_stats_vit.md | cut -c1-200
