#!/bin/bash
# kernel trace of the replayed step graph (one prompt): bash scripts/r05_trace.sh <tag> [extra bench args]
TAG=${1:-r05}; shift || true
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/gpurun_out/prof
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d $ROOT/gpurun_out/prof -o ${TAG}_kt -- python $ROOT/bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-probe --no-shard4 "$@" > $ROOT/gpurun_out/${TAG}_kt.log 2>&1
echo "trace rc=$?"
DB=$(ls $ROOT/gpurun_out/prof/*${TAG}_kt*results.db 2>/dev/null | head -1)
[ -z "$DB" ] && DB=$(find $ROOT/gpurun_out/prof -name "*${TAG}_kt*.db" | head -1)
python $ROOT/tools/rocpd_summary.py "$DB" > $ROOT/gpurun_out/${TAG}_kernel_trace.txt
head -24 $ROOT/gpurun_out/${TAG}_kernel_trace.txt | cut -c1-190
rm -rf $ROOT/gpurun_out/prof
