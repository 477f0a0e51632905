run() { timeout 900 python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', round(d['value'],1), round(d['ms_per_step'],3))"; }
run
run --opt split18=2
run --opt split18=3
run --opt split72=2
run --opt split72=6
run --opt split36=4
run --opt tile_partial=6
run --opt tile_partial=7 --opt split18=2
run --opt tile_f32=5
run --opt tile_f32=7
run --geglu-tile 13
