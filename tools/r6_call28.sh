set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${R6_OUT:-r6_call28}
mkdir -p $O
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
LDS="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS"
i=0
for fl in 32768 0; do
  for set in "$SQ" "$LDS"; do
    i=$((i+1))
    ( PROBE_ROWS=4000 PROBE_KMUL=3 PROBE_FLAGS=$fl timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/sq/pass$i -o p -- python tools/gemm_probe.py 22:qkv 22:wo 22:c_wq 22:w13 22:w2 ) > $O/pmc_pass$i.log 2>&1; echo "pass $i exit=$?"
  done
done
python tools/pmc_sq.py $O/sq > $O/pmc_sq_lds_summary.md 2>$O/pmc_sq_raw.txt; rm -rf $O/sq; cat $O/pmc_sq_lds_summary.md | cut -c1-220; cut -c1-400 $O/pmc_sq_raw.txt
