#!/bin/bash
# same-box A/B of the working tree against _base/ (one prompt x2, four prompts x1): bash scripts/r05_ab.sh <tag>
# _base/ (git-ignored, travels with gpurun) is the library of the commit to compare against, built with
#     rm -rf _base && mkdir _base && git archive HEAD ezaudio_amd tools/ab_prepare.py include | tar -x -C _base && (cd _base && python -m ezaudio_amd.build)
# every line also prints |latents|: equal checksums on both sides = the change is bit-identical
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for i in 1 2; do
  (cd _base && timeout 300 python tools/ab_prepare.py xl 1 --once base 2>&1 | grep -v "Warning\|amdgpu.ids" | sed 's/^/prev /')
  timeout 300 python tools/ab_prepare.py xl 1 --once base 2>&1 | grep -v "Warning\|amdgpu.ids" | sed 's/^/new  /'
done | tee gpurun_out/ab_$1.txt
(cd _base && timeout 300 python tools/ab_prepare.py xl 4 --once base 2>&1 | grep -v "Warning\|amdgpu.ids" | sed 's/^/prev4 /') | tee -a gpurun_out/ab_$1.txt
timeout 300 python tools/ab_prepare.py xl 4 --once base 2>&1 | grep -v "Warning\|amdgpu.ids" | sed 's/^/new4  /' | tee -a gpurun_out/ab_$1.txt
