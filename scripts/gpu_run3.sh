mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --timeout 400 -s > gpurun_out/pytest.txt 2>&1; echo "pytest rc=$?"
tail -4 gpurun_out/pytest.txt
timeout 900 python bench.py --steps 50 --warmup 5 > gpurun_out/bench.txt 2> gpurun_out/bench.err; echo "bench rc=$?"
cat gpurun_out/bench.txt | cut -c1-2500
tail -3 gpurun_out/bench.err
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r1 -- python $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_bench.txt 2> $GRAFT_REPO_ROOT/gpurun_out/prof_bench.err; echo "rocprof rc=$?"
ls -R $GRAFT_REPO_ROOT/gpurun_out/prof | head -20
