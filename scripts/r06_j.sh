#!/bin/bash
# 32-query-tile form of the fused cross-attention (attn_qtile): parity tests, in-process A/B against the 64-row form, stamps of the cross-attention launch
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu.py -m gpu -x -q -k "attention or forward_matches or single_key or odd or smp_xs or placement or per_row" > gpurun_out/r06j_pytest.txt 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r06j_pytest.txt | cut -c1-300
timeout 600 python tools/ab_prepare.py xl 1 base attn_qtile=64 2>&1 | grep -v "Warning\|amdgpu.ids" | tee gpurun_out/r06j_ab1.txt
timeout 600 python tools/ab_prepare.py l 1 base attn_qtile=64 2>&1 | grep -v "Warning\|amdgpu.ids" | tee gpurun_out/r06j_abl.txt
timeout 600 python tools/ab_prepare.py xl 2 base attn_qtile=32 2>&1 | grep -v "Warning\|amdgpu.ids" | tee gpurun_out/r06j_ab2.txt
timeout 600 python tools/ab_prepare.py xl 4 base attn_qtile=32 2>&1 | grep -v "Warning\|amdgpu.ids" | tee gpurun_out/r06j_ab4.txt
echo "== stamps qtile 32"; STAMP_KERNELS='k_attn (cross)' timeout 300 python tools/diag_stamps.py xl 2>&1 | grep -v "Warning\|amdgpu.ids" | tee gpurun_out/r06j_stamps_q32.txt | cut -c1-400
echo "== stamps qtile 64"; EZ_OPTS='attn_qtile=64' STAMP_KERNELS='k_attn (cross)' timeout 300 python tools/diag_stamps.py xl 2>&1 | grep -v "Warning\|amdgpu.ids" | tee gpurun_out/r06j_stamps_q64.txt | cut -c1-400
