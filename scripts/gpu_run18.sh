run() { timeout 900 python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', round(d['value'],1), round(d['ms_per_step'],3))"; }
timeout 900 python -m pytest tests/test_gpu.py -m gpu -q --timeout 400 -k "xs or smp_xs" 2>&1 | tail -1
run
run --prefetch
run
run --prefetch
