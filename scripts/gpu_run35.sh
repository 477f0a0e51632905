#!/bin/bash
timeout 600 python -m pytest tests/test_gpu.py -m gpu -x -q -k "forward or sampler" 2>&1 | grep -v amdgpu.ids | tail -2
timeout 600 python tests/ab_sweep.py xl 1 fuse_qkv=1,0,1,0 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ab15.log
timeout 600 python tests/ab_sweep.py xl 4 fuse_qkv=0,2,0,2 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ab16.log
