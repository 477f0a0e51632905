mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu.py -m gpu -q --timeout 400 -k "gemm or xs or smp_xs or xl" > gpurun_out/pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest.txt
timeout 600 python tests/bench_kernels.py probe 2>&1 | grep -v amdgpu.ids | grep -E "128x64 4x1|128x128 8w r2|128x64 r2" 
timeout 900 python bench.py --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/bench.txt 2> gpurun_out/bench.err; echo "bench rc=$?"
cat gpurun_out/bench.txt | cut -c1-300; tail -3 gpurun_out/bench.err
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r2 -- python $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --no-cpu-baseline > /dev/null 2>&1; echo "rocprof rc=$?"
