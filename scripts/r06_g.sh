#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu.py -m gpu -x -q -k "forward_matches or single_key or odd or smp_xs or per_row or co_resident or placement" > gpurun_out/r06g_pytest.txt 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r06g_pytest.txt | cut -c1-300
timeout 600 python tools/ab_prepare.py xl 1 base qkv_co=2 2>&1 | grep -v "Warning\|amdgpu.ids" | tee gpurun_out/r06g_ab1.txt
timeout 600 python tools/ab_prepare.py xl 4 base qkv_co=2 qkv_co=2+geglu_co=2 2>&1 | grep -v "Warning\|amdgpu.ids" | tee gpurun_out/r06g_ab4.txt
timeout 600 python tools/ab_prepare.py l 1 base qkv_co=2 2>&1 | grep -v "Warning\|amdgpu.ids" | tee gpurun_out/r06g_abl.txt
echo "== stamps new (qkv_co=2)"; EZ_OPTS='qkv_co=2' STAMP_KERNELS='k_gemm (QKV)' timeout 300 python tools/diag_stamps.py xl 2>&1 | grep -v "Warning\|amdgpu.ids" | tee gpurun_out/r06g_stamps_co.txt | cut -c1-400
