mkdir -p gpurun_out
timeout 300 python tests/diag_attn.py 2>&1 | grep -v amdgpu.ids | grep -E "wrong weight|max err" 
timeout 1500 python -m pytest tests -m gpu -q --timeout 400 -s > gpurun_out/pytest.txt 2>&1; echo "pytest rc=$?"
grep -E "rel-L2|passed|failed|FAILED" gpurun_out/pytest.txt | tail -30
timeout 600 python tests/bench_kernels.py all > gpurun_out/bench_kernels.txt 2>&1; echo "bench_kernels rc=$?"
cat gpurun_out/bench_kernels.txt
timeout 900 python bench.py --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/bench.txt 2> gpurun_out/bench.err; echo "bench rc=$?"
cat gpurun_out/bench.txt | cut -c1-400; tail -3 gpurun_out/bench.err
