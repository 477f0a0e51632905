#!/bin/bash
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r9p4 -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --prompts 4 --steps 100 --warmup 50 > $GRAFT_REPO_ROOT/gpurun_out/prof_bench_p4.txt 2>&1; echo "rocprof rc=$?"
tail -1 $GRAFT_REPO_ROOT/gpurun_out/prof_bench_p4.txt | cut -c1-200
