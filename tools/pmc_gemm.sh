#!/bin/bash
# PMC passes over tools/gemm_probe.py (counters in their own runs, kernel-trace only - see MI355X_MICROARCH.md
# "rocprofv3 PMC slots").  Output: gpurun_out/pmc_gemm/pass*/ csv files.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_gemm
mkdir -p $OUT
run() { # name, counters...
  local name=$1; shift
  timeout 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o p -- python tools/gemm_probe.py "${PROBE[@]}" > $OUT/$name.log 2>&1
  echo "$name exit=$?"
}
PROBE=("$@")
run p1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE
run p2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 TCC_HIT_sum TCC_MISS_sum
run p3 FETCH_SIZE
run p4 WRITE_SIZE
ls -R $OUT | head -30
