#!/bin/bash
# round 5, first GPU call: GPU suite of the pruned tree (+ parity log), same-box A/B against the round-4 tree (_base/), bench
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export EZ_PARITY_LOG=$PWD/gpurun_out/parity_gpu1.txt
rm -f "$EZ_PARITY_LOG"
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu1.txt 2>&1
echo "pytest rc=$?"; tail -5 gpurun_out/pytest_gpu1.txt
for i in 1 2; do
  (cd _base && timeout 300 python tools/ab_prepare.py xl 1 --once base 2>&1 | grep -v Warning | sed 's/^/old /')
  timeout 300 python tools/ab_prepare.py xl 1 --once base xkey1=0 2>&1 | grep -v Warning | sed 's/^/new /'
done | tee gpurun_out/ab_gpu1.txt
(cd _base && timeout 300 python tools/ab_prepare.py xl 4 --once base 2>&1 | grep -v Warning | sed 's/^/old4 /') | tee -a gpurun_out/ab_gpu1.txt
timeout 300 python tools/ab_prepare.py xl 4 --once base 2>&1 | grep -v Warning | sed 's/^/new4 /' | tee -a gpurun_out/ab_gpu1.txt
timeout 300 python tools/ab_prepare.py l 1 --once base xkey1=0 2>&1 | grep -v Warning | sed 's/^/newL /' | tee -a gpurun_out/ab_gpu1.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_gpu1.json 2> gpurun_out/bench_gpu1.err
echo "bench rc=$?"; cat gpurun_out/bench_gpu1.json | cut -c1-600
