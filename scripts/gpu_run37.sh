#!/bin/bash
timeout 600 python -m pytest tests/test_gpu.py -m gpu -x -q -s -k "forward or sampler" 2>&1 | grep -v amdgpu.ids | grep -E "s64|^l |passed|failed|Error" | tail -8
timeout 600 python tests/ab_sweep.py l 1 fuse_qkv=1,0,1,0 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ab17.log
