#!/bin/bash
# whole-round same-box A/B: the round-5 final tree (_r05/, built from commit 9ccf7b3's ezaudio_amd) against the working tree; one prompt x3, four prompts x2, EzAudio-L x1
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for i in 1 2 3; do
  (cd _r05 && timeout 300 python tools/ab_prepare.py xl 1 --once base 2>&1 | grep -v "Warning\|amdgpu.ids" | sed 's/^/r05  /')
  timeout 300 python tools/ab_prepare.py xl 1 --once base 2>&1 | grep -v "Warning\|amdgpu.ids" | sed 's/^/r06  /'
done | tee gpurun_out/r06_ab_r05_vs_final.txt
for i in 1 2; do
  (cd _r05 && timeout 300 python tools/ab_prepare.py xl 4 --once base 2>&1 | grep -v "Warning\|amdgpu.ids" | sed 's/^/r05 4p /')
  timeout 300 python tools/ab_prepare.py xl 4 --once base 2>&1 | grep -v "Warning\|amdgpu.ids" | sed 's/^/r06 4p /'
done | tee -a gpurun_out/r06_ab_r05_vs_final.txt
(cd _r05 && timeout 300 python tools/ab_prepare.py l 1 --once base 2>&1 | grep -v "Warning\|amdgpu.ids" | sed 's/^/r05 L  /') | tee -a gpurun_out/r06_ab_r05_vs_final.txt
timeout 300 python tools/ab_prepare.py l 1 --once base 2>&1 | grep -v "Warning\|amdgpu.ids" | sed 's/^/r06 L  /' | tee -a gpurun_out/r06_ab_r05_vs_final.txt
