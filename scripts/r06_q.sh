#!/bin/bash
# cross-attention clipped to the last valid key of the covered batch elements (AttnArgs.Lk_used): parity tests, same-box A/B against _base/, stamps
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu.py tests/test_controlnet.py -m gpu -x -q -k "attention or forward_matches or single_key or odd or smp or placement or context or controlnet or graph" > gpurun_out/r06q_pytest.txt 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r06q_pytest.txt | cut -c1-300
bash scripts/r05_ab.sh r06q
echo "== stamps new"; STAMP_KERNELS='k_attn (cross)' timeout 300 python tools/diag_stamps.py xl 2>&1 | grep -v "Warning\|amdgpu.ids" | tee gpurun_out/r06q_stamps_new.txt | cut -c1-400
echo "== stamps base"; (cd _base && STAMP_KERNELS='k_attn (cross)' timeout 300 python tools/diag_stamps.py xl 2>&1 | grep -v "Warning\|amdgpu.ids" | tee ../gpurun_out/r06q_stamps_base.txt | cut -c1-400)
