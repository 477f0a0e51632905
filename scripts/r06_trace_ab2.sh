#!/bin/bash
# kernel traces of the working tree (with extra bench args, e.g. --opt skip_z=0) and of _base/ (plain) on one box: bash scripts/r06_trace_ab2.sh <tag> [bench args of the NEW side]
TAG=${1:-r06}; shift || true
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
for side in new base; do
  SRC=$ROOT; EXTRA="$@"; [ $side = base ] && SRC=$ROOT/_base && EXTRA=""
  mkdir -p $ROOT/gpurun_out/prof_$side
  (cd /tmp && export TMPDIR=/tmp && cd $SRC && timeout 600 rocprofv3 --kernel-trace -d $ROOT/gpurun_out/prof_$side -o ${TAG}_kt -- python $SRC/bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-probe --no-shard4 $EXTRA > $ROOT/gpurun_out/${TAG}_${side}_kt.log 2>&1)
  DB=$(find $ROOT/gpurun_out/prof_$side -name "*${TAG}_kt*.db" | head -1)
  python $ROOT/tools/rocpd_summary.py "$DB" > $ROOT/gpurun_out/${TAG}_${side}_kernel_trace.txt
  echo "== $side"; head -18 $ROOT/gpurun_out/${TAG}_${side}_kernel_trace.txt | cut -c1-200
  rm -rf $ROOT/gpurun_out/prof_$side
done
