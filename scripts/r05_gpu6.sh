#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for i in 1 2; do
  timeout 300 python tools/ab_prepare.py xl 1 --once base 2>&1 | grep -v "Warning\|amdgpu.ids" | sed 's/^/t70 /'
  EZ_XOUT_TILE=72 timeout 300 python tools/ab_prepare.py xl 1 --once base 2>&1 | grep -v "Warning\|amdgpu.ids" | sed 's/^/t72 /'
done | tee gpurun_out/ab_gpu6.txt
