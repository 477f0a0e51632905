mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -oE "\b(SQ|TCC|TCP|GRBM|TA|TD)_[A-Z0-9_]+" | sort -u | tr '\n' ' ' > $GRAFT_REPO_ROOT/gpurun_out/counters.txt
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU" "GRBM_GUI_ACTIVE GRBM_COUNT TCC_HIT_sum TCC_MISS_sum" ; do
  n=$(echo $set | cut -c1-12 | tr -d ' ')
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d $GRAFT_REPO_ROOT/gpurun_out/pmc -o pmc_$n --output-format csv -- python $GRAFT_REPO_ROOT/tests/probe_one.py gemm 26 1000 9216 1152 > /dev/null 2>&1; echo "pmc $n rc=$?"
done
ls $GRAFT_REPO_ROOT/gpurun_out/pmc | head
