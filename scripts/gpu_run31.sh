#!/bin/bash
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1; echo "tests exit $?" >> gpurun_out/gpu_tests.log; tail -5 gpurun_out/gpu_tests.log
