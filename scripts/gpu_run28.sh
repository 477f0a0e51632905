mkdir -p gpurun_out/pmc3
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_INSTS_VMEM_RD" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA" "GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" "TA_TA_BUSY_sum TD_TD_BUSY_sum TCP_GATE_EN1_sum TCP_TCR_TCP_STALL_CYCLES_sum" "TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum TCC_BUSY_sum" ; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $set -d $GRAFT_REPO_ROOT/gpurun_out/pmc3 -o mlpout_$i --output-format csv -- python $GRAFT_REPO_ROOT/tests/probe_one.py gemm 37 1000 1152 4608 3 > /dev/null 2>&1; echo "pmc $i rc=$?"
done
ls $GRAFT_REPO_ROOT/gpurun_out/pmc3 | head -30
