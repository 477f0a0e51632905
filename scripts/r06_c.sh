#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu.py -m gpu -x -q -k "attention or forward_matches or single_key or odd or smp_xs or per_row" > gpurun_out/r06c_pytest.txt 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r06c_pytest.txt | cut -c1-300
timeout 300 tools/_run/attn_bench 2>&1 | tee gpurun_out/r06c_attn_bench.txt | cut -c1-250
bash scripts/r05_ab.sh r06c
