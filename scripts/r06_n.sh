#!/bin/bash
# pinned / batched kernel arguments (fused-QKV epilogue, k_gemm_ks, k_attn, k_row_w, tail kernels): whole GPU suite, same-box A/B against _base/, QKV stamps, kernel trace
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r06n_pytest.txt 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r06n_pytest.txt | cut -c1-300
bash scripts/r05_ab.sh r06n
echo "== stamps new"; STAMP_KERNELS='k_gemm (QKV)' timeout 300 python tools/diag_stamps.py xl 2>&1 | grep -v "Warning\|amdgpu.ids" | tee gpurun_out/r06n_stamps_new.txt | cut -c1-400
echo "== stamps base"; (cd _base && STAMP_KERNELS='k_gemm (QKV)' timeout 300 python tools/diag_stamps.py xl 2>&1 | grep -v "Warning\|amdgpu.ids" | tee ../gpurun_out/r06n_stamps_base.txt | cut -c1-400)
(cd /tmp && export TMPDIR=/tmp && cd $OLDPWD && timeout 600 rocprofv3 --kernel-trace -d gpurun_out/prof_n -o r06n_kt -- python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-probe --no-shard4 > gpurun_out/r06n_kt.log 2>&1)
DB=$(find gpurun_out/prof_n -name "*r06n_kt*.db" | head -1)
python tools/rocpd_summary.py "$DB" > gpurun_out/r06n_kernel_trace.txt; head -24 gpurun_out/r06n_kernel_trace.txt | cut -c1-200
rm -rf gpurun_out/prof_n
