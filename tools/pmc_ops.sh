#!/bin/bash
# SQ / LDS counters of the non-GEMM DiT kernels at the benchmark shape (tools/op_bench.py), two PMC passes.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_ops
mkdir -p $OUT
run() { local name=$1; shift
  timeout 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o p -- python tools/op_bench.py > $OUT/$name.log 2>&1; echo "$name exit=$?"; }
run p1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES
run p2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_FLAT
