#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_gpu.py -m gpu -q -k "gemm_against or geglu_epilogue or layernorm_algebra or forward_matches or graph_equals or co_resident or per_row" > gpurun_out/r06w_pytest.txt 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|FAILED" gpurun_out/r06w_pytest.txt | cut -c1-250 | head -40
