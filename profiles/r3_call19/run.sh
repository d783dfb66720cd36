#!/bin/bash
# GPU call 19 of round 3: the N-rank bench path on the final code in the only form a 1-GPU box allows - two ranks time-sharing
# cuda:0 (gloo): self-launch and the driver's torchrun form; strong scaling as `value`, the weak figure under other_scaling,
# parity mode on.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3_call19
mkdir -p $O
( timeout 500 python bench.py --gpus 2 --share-gpu --steps 2 --warmup 1 --no-cpu-baseline ) > $O/bench_2ranks_selflaunch.log 2>&1; echo "self-launch exit=$?"; tail -1 $O/bench_2ranks_selflaunch.log | cut -c1-600
( timeout 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --share-gpu --steps 2 --warmup 1 --no-cpu-baseline --no-parity-mode ) > $O/bench_2ranks_torchrun.log 2>&1; echo "torchrun exit=$?"; tail -1 $O/bench_2ranks_torchrun.log | cut -c1-600
