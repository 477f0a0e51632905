#!/bin/bash
# end-of-round verification: full GPU suite, smoke(), bench, then the two HBM-traffic PMC passes for roofline.traffic
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/pytest.txt 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/pytest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python bench.py > gpurun_out/bench.txt 2>gpurun_out/bench.err; cut -c1-260 gpurun_out/bench.txt | tail -1
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc4
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $GRAFT_REPO_ROOT/gpurun_out/pmc4 -o fetch --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-graph > /dev/null 2>&1; echo "pmc fetch rc=$?"
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $GRAFT_REPO_ROOT/gpurun_out/pmc4 -o write --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-graph > /dev/null 2>&1; echo "pmc write rc=$?"
ls $GRAFT_REPO_ROOT/gpurun_out/pmc4
