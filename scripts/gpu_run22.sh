#!/bin/bash
# VAE decoder first GPU run: VAE tests verbose, then the full GPU suite (GEMM addressing changed)
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_vae.py -m gpu -x -q -s > gpurun_out/vae_tests.log 2>&1
echo "vae exit $?" >> gpurun_out/vae_tests.log
tail -15 gpurun_out/vae_tests.log
true
echo "all exit $?" >> gpurun_out/gpu_tests.log
tail -5 gpurun_out/gpu_tests.log
