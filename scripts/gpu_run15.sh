mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu.py -m gpu -q --timeout 400 -k "xs or smp or xl" 2>&1 | tail -2
for i in 1 2; do
timeout 900 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('prefetch ON ', d['value'], d['ms_per_step'])"
timeout 900 python bench.py --no-cpu-baseline --no-prefetch 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('prefetch OFF', d['value'], d['ms_per_step'])"
done
