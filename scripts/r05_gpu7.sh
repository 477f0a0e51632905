#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu.py -m gpu -q -k "attention or golden or single_key or odd_head or batch_equals" > gpurun_out/pytest_gpu7.txt 2>&1
echo "pytest rc=$?"; tail -6 gpurun_out/pytest_gpu7.txt | cut -c1-250
for i in 1 2; do
  (cd _base && timeout 300 python tools/ab_prepare.py xl 1 --once base 2>&1 | grep -v "Warning\|amdgpu.ids" | sed 's/^/prev /')
  timeout 300 python tools/ab_prepare.py xl 1 --once base 2>&1 | grep -v "Warning\|amdgpu.ids" | sed 's/^/new  /'
done | tee gpurun_out/ab_gpu7.txt
(cd _base && timeout 300 python tools/ab_prepare.py xl 4 --once base 2>&1 | grep -v "Warning\|amdgpu.ids" | sed 's/^/prev4 /') | tee -a gpurun_out/ab_gpu7.txt
timeout 300 python tools/ab_prepare.py xl 4 --once base 2>&1 | grep -v "Warning\|amdgpu.ids" | sed 's/^/new4  /' | tee -a gpurun_out/ab_gpu7.txt
