#!/bin/bash
# same-box A/B of the working tree against _base/ (one prompt x2, four prompts x1): bash scripts/r05_ab.sh <tag>
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for i in 1 2; do
  (cd _base && timeout 300 python tools/ab_prepare.py xl 1 --once base 2>&1 | grep -v "Warning\|amdgpu.ids" | sed 's/^/prev /')
  timeout 300 python tools/ab_prepare.py xl 1 --once base 2>&1 | grep -v "Warning\|amdgpu.ids" | sed 's/^/new  /'
done | tee gpurun_out/ab_$1.txt
(cd _base && timeout 300 python tools/ab_prepare.py xl 4 --once base 2>&1 | grep -v "Warning\|amdgpu.ids" | sed 's/^/prev4 /') | tee -a gpurun_out/ab_$1.txt
timeout 300 python tools/ab_prepare.py xl 4 --once base 2>&1 | grep -v "Warning\|amdgpu.ids" | sed 's/^/new4  /' | tee -a gpurun_out/ab_$1.txt
