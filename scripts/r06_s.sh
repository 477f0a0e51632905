#!/bin/bash
# LayerNorm-algebra consumers: row statistics / G' / C' by blind (inline-asm) loads and LDS stores, so that the K loop starts behind the FIRST prologue tile: whole GPU suite,
# same-box A/B against _base/ (HEAD), stamps, kernel trace
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r06s_pytest.txt 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r06s_pytest.txt | cut -c1-300
bash scripts/r05_ab.sh r06s
echo "== stamps new"; STAMP_KERNELS='k_gemm (QKV);k_gemm (GEGLU)' timeout 300 python tools/diag_stamps.py xl 2>&1 | grep -v "Warning\|amdgpu.ids" | tee gpurun_out/r06s_stamps_new.txt | cut -c1-300
