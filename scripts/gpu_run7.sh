mkdir -p gpurun_out
timeout 600 python tests/bench_kernels.py probe > gpurun_out/probe_gemm.txt 2>&1; echo "probe rc=$?"
cat gpurun_out/probe_gemm.txt
timeout 900 python -m pytest tests/test_gpu.py -m gpu -q --timeout 400 -k "gemm or xs or smp_xs" > gpurun_out/pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest.txt
timeout 900 python bench.py --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/bench.txt 2> gpurun_out/bench.err; echo "bench rc=$?"
cat gpurun_out/bench.txt | cut -c1-300; tail -3 gpurun_out/bench.err
