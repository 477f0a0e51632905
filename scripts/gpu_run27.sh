#!/bin/bash
mkdir -p gpurun_out
timeout 600 python tests/ab_sweep.py xl 4 tile_partial_big=5,9,4,7,8,10,22,24 tile_f32_big=7,9,8,10,22,27,24,14 geglu_big=2,12,13,7,9,8,10,11,28 split_big=0,2 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ab8.log
timeout 600 python tests/ab_sweep.py l 1 tile_partial=9,4,5,24 tile_f32=25,14,24 tile_qkv=9,-1,22 geglu_tile=-1,12,9,7 split18=3,2,4 split72=3,2,4 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ab9.log
