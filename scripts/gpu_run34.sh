#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu.py -m gpu -x -q -k "gemm or geglu" 2>&1 | grep -v amdgpu.ids | tail -3
timeout 500 python tests/bench_cold.py proj mlpout qkv geglu skip 2>&1 | grep -v amdgpu.ids | cut -c1-400 | tee gpurun_out/cold8.log
timeout 600 python tests/ab_sweep.py xl 1 xcd_map=1 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ab14.log
