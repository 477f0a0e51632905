#!/bin/bash
# LayerNorm-algebra consumers: (mu, r) per LANE, statistics requested in the K loop's tail and merged behind the loop -- nothing of the algebra in front of the loop:
# whole GPU suite, same-box A/B against _base/ (HEAD: the per-workgroup merge in front of the loop), stamps
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r06v_pytest.txt 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r06v_pytest.txt | cut -c1-300
bash scripts/r05_ab.sh r06v
echo "== stamps new"; STAMP_KERNELS='k_gemm (QKV);k_gemm (GEGLU)' timeout 300 python tools/diag_stamps.py xl 2>&1 | grep -v "Warning\|amdgpu.ids" | tee gpurun_out/r06v_stamps_new.txt | cut -c1-300
