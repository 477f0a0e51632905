#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu.py -m gpu -x -q -k "gemm_against or geglu_epilogue or residual_gemm or attention or forward_matches or single_key or odd or smp_xs or per_row or co_resident" > gpurun_out/r06h_pytest.txt 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r06h_pytest.txt | cut -c1-300
bash scripts/r05_ab.sh r06h
