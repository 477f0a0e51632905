#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu.py -m gpu -x -q -k "forward_matches or layernorm_algebra or single_key or odd or smp_xs or per_row or residual_gemm" > gpurun_out/r06e_pytest.txt 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r06e_pytest.txt | cut -c1-300
bash scripts/r05_ab.sh r06e
echo "== stamps new"; STAMP_KERNELS='k_gemm (QKV)' timeout 300 python tools/diag_stamps.py xl 2>&1 | grep -v "Warning\|amdgpu.ids" | tee gpurun_out/r06e_stamps_new.txt | cut -c1-260
echo "== stamps base"; (cd _base && STAMP_KERNELS='k_gemm (QKV)' timeout 300 python tools/diag_stamps.py xl 2>&1 | grep -v "Warning\|amdgpu.ids" | tee ../gpurun_out/r06e_stamps_base.txt | cut -c1-260)
