#!/bin/bash
# split first tile in the 8-wave attention kernel (K half scored while the V half is in flight): parity tests, same-box A/B against _base/ (HEAD), stamps
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu.py -m gpu -x -q -k "attention or forward_matches or single_key or odd or smp_xs or placement" > gpurun_out/r06l_pytest.txt 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r06l_pytest.txt | cut -c1-300
bash scripts/r05_ab.sh r06l
echo "== stamps new"; STAMP_KERNELS='k_attn (self);k_attn (cross)' timeout 300 python tools/diag_stamps.py xl 2>&1 | grep -v "Warning\|amdgpu.ids" | tee gpurun_out/r06l_stamps_new.txt | cut -c1-400
echo "== stamps base"; (cd _base && STAMP_KERNELS='k_attn (self);k_attn (cross)' timeout 300 python tools/diag_stamps.py xl 2>&1 | grep -v "Warning\|amdgpu.ids" | tee ../gpurun_out/r06l_stamps_base.txt | cut -c1-400)
