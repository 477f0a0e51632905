#!/bin/bash
# final Conv1d reading a transposed weight copy (coalesced): parity tests, same-box A/B against _base/, kernel trace
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu.py -m gpu -x -q -k "forward_matches or smp_xs or smp_s or reference_style or graph" > gpurun_out/r06m_pytest.txt 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r06m_pytest.txt | cut -c1-300
bash scripts/r05_ab.sh r06m
(cd /tmp && export TMPDIR=/tmp && cd $OLDPWD && timeout 600 rocprofv3 --kernel-trace -d gpurun_out/prof_m -o r06m_kt -- python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-probe --no-shard4 > gpurun_out/r06m_kt.log 2>&1)
DB=$(find gpurun_out/prof_m -name "*r06m_kt*.db" | head -1)
python tools/rocpd_summary.py "$DB" > gpurun_out/r06m_kernel_trace.txt; head -24 gpurun_out/r06m_kernel_trace.txt | cut -c1-200
rm -rf gpurun_out/prof_m
