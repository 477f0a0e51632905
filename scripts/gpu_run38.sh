#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/pytest.txt 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/pytest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 900 python bench.py > gpurun_out/bench.txt 2>gpurun_out/bench.err; cut -c1-330 gpurun_out/bench.txt | tail -1
timeout 300 python bench.py --no-cpu-baseline --size l 2>/dev/null | cut -c1-200
