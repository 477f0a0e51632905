#!/bin/bash
# timing bound: the LayerNorm-algebra consumers WITHOUT z_finish (_hack/, wrong results): what their prologue costs beyond the plain kernel's
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for i in 1 2; do
  timeout 300 python tools/ab_prepare.py xl 1 --once base 2>&1 | grep -v "Warning\|amdgpu.ids" | sed 's/^/base    /'
  (cd _hack && timeout 300 python tools/ab_prepare.py xl 1 --once base 2>&1 | grep -v "Warning\|amdgpu.ids" | sed 's/^/nozfin  /')
done | tee gpurun_out/r06r_ab.txt
echo "== stamps nozfin"; (cd _hack && STAMP_KERNELS='k_gemm (QKV);k_gemm (GEGLU)' timeout 300 python tools/diag_stamps.py xl 2>&1 | grep -v "Warning\|amdgpu.ids" | tee ../gpurun_out/r06r_stamps_hack.txt | cut -c1-300)
