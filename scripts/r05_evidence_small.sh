#!/bin/bash
# reduced evidence run (PMC passes + driver-form bench + one-prompt kernel trace) for the tree as it is: bash scripts/r05_evidence_small.sh <tag>
TAG=${1:-r05g}
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ROOT=$PWD
mkdir -p gpurun_out
PMC_STEPS=8 PMC_WARM=2 bash scripts/pmc_step.sh ${TAG} 2>&1 | tail -3
cd $ROOT
cp gpurun_out/${TAG}_pmc_step.json profiles/${TAG}_pmc_step.json   # so that the bench line below quotes it (src_hash match)
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_xl_driverform.json 2> gpurun_out/${TAG}_bench.err; echo "bench driverform rc=$?"
python - <<PY
import json
d=json.load(open('gpurun_out/${TAG}_bench_xl_driverform.json'))
print(round(d['value'],1), round(d['ms_per_step'],3), 'frac', round(d['roofline']['frac'],4), 'traffic', d['roofline']['traffic'], 'shard4', (d.get('config4_shard') or {}).get('ms_per_step'), 'cpu', d.get('cpu_baseline',{}).get('value'))
PY
bash scripts/r05_trace.sh ${TAG} > gpurun_out/${TAG}_trace.log 2>&1; echo "trace rc=$?"
