#!/bin/bash
# k_gemm_ks epilogue sub-phases for every kind of un-split residual launch + the tail GEMMs, then a kernel trace of the working tree
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
STAMP_KERNELS='k_gemm (un-split residual: attention-out DUAL);k_gemm (un-split residual: cross-out);k_gemm (un-split residual: MLP-out);k_gemm (un-split residual: skip_linear)' timeout 600 python tools/diag_stamps.py xl 2>&1 | grep -v "Warning\|amdgpu.ids" | tee gpurun_out/r06k_stamps_ks.txt | cut -c1-500
EZ_OPTS='xkey1=0' STAMP_KERNELS='k_gemm (un-split residual: attention-out);k_gemm (un-split residual: cross-out)' timeout 600 python tools/diag_stamps.py xl 2>&1 | grep -v "Warning\|amdgpu.ids" | tee gpurun_out/r06k_stamps_ks_plain.txt | cut -c1-500
(cd /tmp && export TMPDIR=/tmp && cd $OLDPWD && timeout 600 rocprofv3 --kernel-trace -d gpurun_out/prof_k -o r06k_kt -- python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-probe --no-shard4 > gpurun_out/r06k_kt.log 2>&1)
DB=$(find gpurun_out/prof_k -name "*r06k_kt*.db" | head -1)
python tools/rocpd_summary.py "$DB" > gpurun_out/r06k_kernel_trace.txt; head -24 gpurun_out/r06k_kernel_trace.txt | cut -c1-200
rm -rf gpurun_out/prof_k
