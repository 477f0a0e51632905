#!/bin/bash
# HBM-traffic and MFMA-utilisation counters of ONE denoising step, per kernel (runs on the MI355X box from the repo root):
#     bash scripts/pmc_step.sh [tag] [extra bench.py args]
# Six separate rocprofv3 --pmc passes of the SAME bench command (FETCH_SIZE and WRITE_SIZE do not fit one pass, MI355X_MICROARCH.md
# "rocprofv3 PMC slots"), eager launches so that every dispatch is attributed to its kernel, then tools/pmc_step_summary.py folds
# the CSVs into gpurun_out/<tag>_pmc_step.json (copy it to profiles/).  The JSON records the hash of the kernel sources it was
# measured on; bench.py quotes `roofline.traffic` from it only while that hash still matches the tree.
TAG=${1:-r02}
shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
STEPS=${PMC_STEPS:-20}
WARM=${PMC_WARM:-5}
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps $STEPS --warmup $WARM --no-cpu-baseline --no-probe --no-shard4 --no-graph $*"
pass() {
  name=$1; shift
  timeout 900 rocprofv3 --kernel-trace --pmc "$@" -d "$OUT" -o "$name" --output-format csv -- $BENCH > "$OUT/$name.log" 2>&1
  echo "pmc pass $name rc=$?"
}
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass sq SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA GRBM_GUI_ACTIVE
# LDS pipe and L2: is the K loop waiting on LDS (bank conflicts, instruction issue) or on memory?  
pass lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_WAVE_CYCLES SQ_INSTS_VALU
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
# how much of the L2<->fabric traffic reaches DRAM (the rest is served by the Infinity Cache)
pass dram TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_DRAM_sum TCC_EA0_WRREQ_sum TCC_BUBBLE_sum
python "$ROOT/tools/pmc_step_summary.py" "$OUT" $((STEPS + WARM)) --tag "$TAG" --args "$*" > "$ROOT/gpurun_out/${TAG}_pmc_step.json"
python - <<PY
import json
d = json.load(open("$ROOT/gpurun_out/${TAG}_pmc_step.json"))
print('traffic per step: fetch %.2f GB (corrected) + write %.2f GB = %.2f GB; MFMA busy (step) %.1f %%' % (
    d['fetch_bytes_per_step'] / 1e9, d['write_bytes_per_step'] / 1e9, d['traffic_bytes_per_step'] / 1e9, 100 * d.get('mfma_busy_frac', 0)))
PY
