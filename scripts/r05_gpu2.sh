#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export EZ_PARITY_LOG=$PWD/gpurun_out/parity_gpu3.txt
rm -f "$EZ_PARITY_LOG"
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu3.txt 2>&1
echo "pytest rc=$?"; tail -40 gpurun_out/pytest_gpu3.txt | cut -c1-250
timeout 300 python tools/ab_prepare.py xl 1 base xkey1=0 2>&1 | grep -v "Warning\|amdgpu.ids" | tee gpurun_out/ab_gpu3.txt
