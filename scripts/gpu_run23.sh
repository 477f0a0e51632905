#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu.py -m gpu -x -q -k "gemm or forward" > gpurun_out/gpu_tests.log 2>&1; echo "tests exit $?" >> gpurun_out/gpu_tests.log; tail -4 gpurun_out/gpu_tests.log
timeout 600 python tests/ab_sweep.py xl 1 xcd_map=1,0 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ab2.log
