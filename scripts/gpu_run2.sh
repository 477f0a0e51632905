mkdir -p gpurun_out
timeout 600 python tests/diag_sampler.py > gpurun_out/diag_sampler.txt 2>&1; echo "diag sampler rc=$?"
timeout 600 python -m pytest tests/test_gpu.py -m gpu -q --timeout 400 -k "geglu or attention" > gpurun_out/pytest2.txt 2>&1; echo "pytest rc=$?"
tail -5 gpurun_out/pytest2.txt
