cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r4 -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > /dev/null 2>&1; echo "rocprof rc=$?"
