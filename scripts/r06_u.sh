#!/bin/bash
# as r06_t + the row statistics requested in FRONT of the first LDS-DMA: GPU tests (subset), same-box A/B against _base/ (HEAD), stamps
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_gpu.py -m gpu -x -q -k "forward_matches or layernorm_algebra or single_key or odd or smp_xs or smp_s or per_row or co_resident or gemm_against or geglu_epilogue or residual_gemm or graph" > gpurun_out/r06u_pytest.txt 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r06u_pytest.txt | cut -c1-300
bash scripts/r05_ab.sh r06u
echo "== stamps new"; STAMP_KERNELS='k_gemm (QKV);k_gemm (GEGLU)' timeout 300 python tools/diag_stamps.py xl 2>&1 | grep -v "Warning\|amdgpu.ids" | tee gpurun_out/r06u_stamps_new.txt | cut -c1-300
