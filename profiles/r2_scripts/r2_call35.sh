#!/bin/bash
# Round 2, GPU call 35: SQ counters of the shipped GEMM kernels on the DiT shapes (tools/pmc_gemm.sh passes 1 and 2 over
# tools/gemm_probe.py: 8-phase 256x256 on qkv / w13 / c_wq / w2 at M = 8000) -> MFMA busy, wait / stall split, LDS conflicts.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_gemm
mkdir -p $OUT
run() { local name=$1; shift
  timeout 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o p -- python tools/gemm_probe.py 22:qkv 22:w13 22:c_wq 22:w2 > $OUT/$name.log 2>&1; echo "$name exit=$?"; }
run p1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE
run p2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES GRBM_GUI_ACTIVE
mkdir -p gpurun_out/r2_call35
python tools/pmc_sq.py $OUT > gpurun_out/r2_call35/pmc_gemm8.md 2> gpurun_out/r2_call35/pmc_sq.err; cat gpurun_out/r2_call35/pmc_gemm8.md; tail -3 gpurun_out/r2_call35/pmc_sq.err
rm -rf $OUT
