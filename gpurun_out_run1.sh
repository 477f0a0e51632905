mkdir -p gpurun_out
rocminfo | grep -E "Marketing|gfx" | head -4 > gpurun_out/sys.txt; nproc >> gpurun_out/sys.txt; lscpu | grep "Model name" >> gpurun_out/sys.txt
timeout 300 python tests/diag_forward.py xs 96 > gpurun_out/diag_xs.txt 2>&1; echo "diag xs rc=$?"
timeout 300 python tests/diag_forward.py xs64 96 > gpurun_out/diag_xs64.txt 2>&1; echo "diag xs64 rc=$?"
timeout 1500 python -m pytest tests -m gpu -q --timeout 400 -s > gpurun_out/pytest.txt 2>&1; echo "pytest rc=$?"
tail -5 gpurun_out/pytest.txt
timeout 600 python bench.py --steps 20 --warmup 3 > gpurun_out/bench.txt 2> gpurun_out/bench.err; echo "bench rc=$?"
cat gpurun_out/bench.txt | cut -c1-1500
