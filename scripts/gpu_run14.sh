mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --timeout 400 > gpurun_out/pytest.txt 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/pytest.txt
timeout 900 python bench.py > gpurun_out/bench.txt 2> gpurun_out/bench.err; echo "bench rc=$?"
cat gpurun_out/bench.txt | cut -c1-3000; tail -3 gpurun_out/bench.err
timeout 900 python bench.py --prompts 4 --no-cpu-baseline > gpurun_out/bench_p4.txt 2>> gpurun_out/bench.err; echo "bench p4 rc=$?"
cat gpurun_out/bench_p4.txt | cut -c1-700
timeout 900 python bench.py --size l --no-cpu-baseline > gpurun_out/bench_l.txt 2>> gpurun_out/bench.err; echo "bench L rc=$?"
cat gpurun_out/bench_l.txt | cut -c1-400
python __graft_entry__.py smoke 2>&1 | tail -4
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r3 -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > /dev/null 2>&1; echo "rocprof rc=$?"
