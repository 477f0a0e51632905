#!/bin/bash
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r8 -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_bench.txt 2>&1; echo "rocprof rc=$?"
cd $GRAFT_REPO_ROOT
timeout 300 python bench.py --no-cpu-baseline --prompts 4 > gpurun_out/bench_p4.txt 2>/dev/null; tail -1 gpurun_out/bench_p4.txt | cut -c1-200
timeout 300 python bench.py --no-cpu-baseline --size l > gpurun_out/bench_l.txt 2>/dev/null; tail -1 gpurun_out/bench_l.txt | cut -c1-200
timeout 300 python bench.py --no-cpu-baseline --controlnet > gpurun_out/bench_cn.txt 2>/dev/null; tail -1 gpurun_out/bench_cn.txt | cut -c1-200
