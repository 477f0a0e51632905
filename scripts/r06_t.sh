#!/bin/bash
# LayerNorm-algebra consumers / producers: row statistics by blind loads, per-column vectors by LDS-DMA straight into their LDS table, nothing of it waited for beyond the first K tile:
# whole GPU suite, same-box A/B against _base/ (HEAD), stamps
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r06t_pytest.txt 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r06t_pytest.txt | cut -c1-300
bash scripts/r05_ab.sh r06t
echo "== stamps new"; STAMP_KERNELS='k_gemm (QKV);k_gemm (GEGLU)' timeout 300 python tools/diag_stamps.py xl 2>&1 | grep -v "Warning\|amdgpu.ids" | tee gpurun_out/r06t_stamps_new.txt | cut -c1-300
