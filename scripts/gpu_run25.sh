#!/bin/bash
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1; echo "tests exit $?" >> gpurun_out/gpu_tests.log; tail -4 gpurun_out/gpu_tests.log
timeout 600 python bench.py > gpurun_out/bench.txt 2> gpurun_out/bench.err; tail -1 gpurun_out/bench.txt | cut -c1-1500
