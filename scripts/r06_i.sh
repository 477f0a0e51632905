#!/bin/bash
# attention prologue with batched kernel arguments: attention / forward tests, then same-box A/B against _base/ (= HEAD 19f91a7)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu.py -m gpu -x -q -k "attention or forward_matches or single_key or odd or smp_xs" > gpurun_out/r06i_pytest.txt 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r06i_pytest.txt | cut -c1-300
bash scripts/r05_ab.sh r06i
