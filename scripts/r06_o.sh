#!/bin/bash
# timing experiments on the LayerNorm-algebra consumers' prologue: _hack2/ = row statistics requested in FRONT of the first LDS-DMA (valid results);
# _hack/ = that + G' / C' of slot 0 without waiting for the step counter (TIMING ONLY: wrong results)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for i in 1 2; do
  timeout 300 python tools/ab_prepare.py xl 1 --once base 2>&1 | grep -v "Warning\|amdgpu.ids" | sed 's/^/base   /'
  (cd _hack2 && timeout 300 python tools/ab_prepare.py xl 1 --once base 2>&1 | grep -v "Warning\|amdgpu.ids" | sed 's/^/zearly /')
  (cd _hack && timeout 300 python tools/ab_prepare.py xl 1 --once base 2>&1 | grep -v "Warning\|amdgpu.ids" | sed 's/^/zearly+slot0 /')
done | tee gpurun_out/r06o_ab.txt
echo "== stamps base"; STAMP_KERNELS='k_gemm (QKV);k_gemm (GEGLU)' timeout 300 python tools/diag_stamps.py xl 2>&1 | grep -v "Warning\|amdgpu.ids" | tee gpurun_out/r06o_stamps_base.txt | cut -c1-300
echo "== stamps zearly"; (cd _hack2 && STAMP_KERNELS='k_gemm (QKV);k_gemm (GEGLU)' timeout 300 python tools/diag_stamps.py xl 2>&1 | grep -v "Warning\|amdgpu.ids" | tee ../gpurun_out/r06o_stamps_zearly.txt | cut -c1-300)
echo "== stamps zearly+slot0"; (cd _hack && STAMP_KERNELS='k_gemm (QKV);k_gemm (GEGLU)' timeout 300 python tools/diag_stamps.py xl 2>&1 | grep -v "Warning\|amdgpu.ids" | tee ../gpurun_out/r06o_stamps_hack.txt | cut -c1-300)
