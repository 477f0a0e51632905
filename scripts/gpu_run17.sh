mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --timeout 400 > gpurun_out/pytest.txt 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/pytest.txt
timeout 900 python bench.py --no-cpu-baseline 2>/dev/null | cut -c1-200
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $GRAFT_REPO_ROOT/gpurun_out/pmc2 -o fetch --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-graph > /dev/null 2>&1; echo "pmc fetch rc=$?"
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $GRAFT_REPO_ROOT/gpurun_out/pmc2 -o write --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-graph > /dev/null 2>&1; echo "pmc write rc=$?"
ls $GRAFT_REPO_ROOT/gpurun_out/pmc2
