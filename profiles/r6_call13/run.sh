set -x
O=gpurun_out/${R6_OUT:-r6_call13}
mkdir -p $O
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_x3_gpu.py -m gpu -q -s -p no:cacheprovider -k "split_on_the_fly" > $O/tests_fly.log 2>&1
grep "passed\|failed\|Error" $O/tests_fly.log | cut -c1-260
timeout 600 python tools/fly_probe.py > $O/fly_probe.log 2>&1; cat $O/fly_probe.log | cut -c1-400
