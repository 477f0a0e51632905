#!/bin/bash
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r5 -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_bench.txt 2>&1; echo "rocprof rc=$?"
tail -1 $GRAFT_REPO_ROOT/gpurun_out/prof_bench.txt | cut -c1-400
