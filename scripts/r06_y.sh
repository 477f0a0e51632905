#!/bin/bash
# the producers' statistics store as a write-through store (it was the one plain store of k_gemm_ks): quick tests + same-box A/B against _base/ (HEAD)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu.py -m gpu -x -q -k "residual_gemm or forward_matches or layernorm_algebra or smp_xs" > gpurun_out/r06y_pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r06y_pytest.txt | cut -c1-200
bash scripts/r05_ab.sh r06y
