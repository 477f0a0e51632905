#!/bin/bash
# round 6, first GPU call: tr_b16 probe, unit tests of the co-resident kernel, kernel-level A/B, in-situ A/B of geglu_co
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
tools/_run/tr_probe > gpurun_out/r06a_tr_probe.txt 2>&1; echo "tr_probe rc=$?"
timeout 900 python -m pytest tests/test_gpu.py -m gpu -x -q -k "gemm_against or geglu_epilogue or context_with_another or single_key" > gpurun_out/r06a_pytest.txt 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r06a_pytest.txt | cut -c1-300
for sh in geglu geglu_b4 qkv_b4; do timeout 300 tools/_run/gemm_bench $sh; done > gpurun_out/r06a_gemm_bench.txt 2>&1; echo "gemm_bench rc=$?"
grep -v "abl\|old\|62" gpurun_out/r06a_gemm_bench.txt | cut -c1-220
timeout 600 python tools/ab_prepare.py xl 1 base geglu_co=2 2>&1 | grep -v "Warning\|amdgpu.ids" | tee gpurun_out/r06a_ab1.txt
timeout 600 python tools/ab_prepare.py xl 4 base geglu_co=0 2>&1 | grep -v "Warning\|amdgpu.ids" | tee gpurun_out/r06a_ab4.txt
