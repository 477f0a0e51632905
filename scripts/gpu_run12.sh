timeout 900 python -m pytest tests/test_gpu.py -m gpu -q --timeout 400 -k "gemm" 2>&1 | tail -2
timeout 600 python tests/bench_kernels.py gemm 2>&1 | grep -v amdgpu.ids | grep -v worst | grep -E "geglu|qkv |mlp-out |proj " | cut -c1-420
