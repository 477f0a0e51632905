#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export EZ_PARITY_LOG=$PWD/gpurun_out/parity_gpu4.txt
rm -f "$EZ_PARITY_LOG"
timeout 600 python -m pytest tests/test_gpu.py -m gpu -q -k "single_key or residual_gemm or xl_b8 or batch_equals" > gpurun_out/pytest_gpu4.txt 2>&1
echo "pytest rc=$?"; tail -12 gpurun_out/pytest_gpu4.txt | cut -c1-250
timeout 300 python tools/ab_prepare.py xl 4 base xkey1=0 2>&1 | grep -v "Warning\|amdgpu.ids" | tee gpurun_out/ab_gpu4.txt
timeout 300 python tools/ab_prepare.py xl 2 --once base xkey1=0 2>&1 | grep -v "Warning\|amdgpu.ids" | tee -a gpurun_out/ab_gpu4.txt
