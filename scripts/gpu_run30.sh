#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu.py tests/test_controlnet.py -m gpu -s -x -q -k "forward or sampler or controlnet" 2>&1 | grep -v amdgpu.ids > gpurun_out/gpu_tests.log; tail -6 gpurun_out/gpu_tests.log
timeout 600 python tests/ab_sweep.py xl 1 fuse_qkv=1,0,1,0 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ab10.log
