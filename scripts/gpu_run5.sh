mkdir -p gpurun_out
timeout 300 python tests/diag_attn.py > gpurun_out/diag_attn.txt 2>&1; echo rc=$?
cat gpurun_out/diag_attn.txt
