#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu.py -m gpu -x -q -k "forward_matches or layernorm_algebra or single_key or odd or smp_xs or smp_l or per_row or placement" > gpurun_out/r06d_pytest.txt 2>&1; echo "pytest rc=$?"; tail -8 gpurun_out/r06d_pytest.txt | cut -c1-300
bash scripts/r05_ab.sh r06d
