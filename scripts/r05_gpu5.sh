#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export EZ_PARITY_LOG=$PWD/gpurun_out/parity_gpu5.txt
rm -f "$EZ_PARITY_LOG"
timeout 900 python -m pytest tests/test_gpu.py -m gpu -q -k "geglu or golden or odd_lengths or qkv or placement" > gpurun_out/pytest_gpu5.txt 2>&1
echo "pytest rc=$?"; tail -6 gpurun_out/pytest_gpu5.txt | cut -c1-250
for i in 1 2; do
  (cd _base && timeout 300 python tools/ab_prepare.py xl 1 --once base 2>&1 | grep -v "Warning\|amdgpu.ids" | sed 's/^/prev /')
  timeout 300 python tools/ab_prepare.py xl 1 --once base 2>&1 | grep -v "Warning\|amdgpu.ids" | sed 's/^/new  /'
done | tee gpurun_out/ab_gpu5.txt
(cd _base && timeout 300 python tools/ab_prepare.py xl 4 --once base fuse_q2=2 2>&1 | grep -v "Warning\|amdgpu.ids" | sed 's/^/prev4 /') | tee -a gpurun_out/ab_gpu5.txt
timeout 300 python tools/ab_prepare.py xl 4 --once base fuse_q2=2 2>&1 | grep -v "Warning\|amdgpu.ids" | sed 's/^/new4  /' | tee -a gpurun_out/ab_gpu5.txt
