mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu.py -m gpu -q --timeout 400 -k "gemm or xs or xl" 2>&1 | tail -2
timeout 600 python tests/bench_kernels.py gemm 2>&1 | grep -v amdgpu.ids | grep -v worst | cut -c1-330
timeout 900 python bench.py --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/bench.txt 2> gpurun_out/bench.err; echo "bench rc=$?"
cat gpurun_out/bench.txt | cut -c1-300; tail -3 gpurun_out/bench.err
