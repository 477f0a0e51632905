#!/bin/bash
# round 6: evidence run on the final tree (GPU suite + parity log, smoke, bench lines, kernel traces, PMC passes): bash scripts/r06_final.sh <tag>
TAG=${1:-r06p}
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ROOT=$PWD
mkdir -p gpurun_out
export EZ_PARITY_LOG=$ROOT/gpurun_out/${TAG}_parity.txt
rm -f "$EZ_PARITY_LOG"
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/${TAG}_pytest_gpu.txt 2>&1
echo "pytest rc=$?"; tail -4 gpurun_out/${TAG}_pytest_gpu.txt | cut -c1-200
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_smoke.txt 2>&1; echo "smoke rc=$?"
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_xl_driverform.json 2> gpurun_out/${TAG}_bench.err; echo "bench driverform rc=$?"
timeout 600 python bench.py --no-cpu-baseline --no-shard4 > gpurun_out/${TAG}_bench_xl.json 2>> gpurun_out/${TAG}_bench.err; echo "bench xl rc=$?"
timeout 600 python bench.py --no-cpu-baseline --no-probe --prompts 4 > gpurun_out/${TAG}_bench_xl_4prompts.json 2>> gpurun_out/${TAG}_bench.err; echo "bench 4p rc=$?"
timeout 600 python bench.py --no-cpu-baseline --no-probe --size l > gpurun_out/${TAG}_bench_l.json 2>> gpurun_out/${TAG}_bench.err; echo "bench l rc=$?"
timeout 600 python bench.py --no-cpu-baseline --no-probe --controlnet > gpurun_out/${TAG}_bench_xl_controlnet.json 2>> gpurun_out/${TAG}_bench.err; echo "bench cn rc=$?"
for f in xl_driverform xl xl_4prompts l xl_controlnet; do python - <<PY
import json
d=json.load(open('gpurun_out/${TAG}_bench_$f.json'))
print('$f', round(d['value'],1), d['unit'], round(d['ms_per_step'],3),'ms', 'frac', round(d['roofline']['frac'],4), 'shard4', (d.get('config4_shard') or {}).get('ms_per_step'))
PY
done
bash scripts/r05_trace.sh ${TAG} > gpurun_out/${TAG}_trace.log 2>&1; echo "trace rc=$?"
bash scripts/r05_trace.sh ${TAG}_4p --prompts 4 > gpurun_out/${TAG}_4p_trace.log 2>&1; echo "trace4 rc=$?"
cd $ROOT
PMC_STEPS=8 PMC_WARM=2 bash scripts/pmc_step.sh ${TAG} 2>&1 | tail -8
