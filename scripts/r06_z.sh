#!/bin/bash
# second look at the write-through statistics store (3 alternations against _base/ = HEAD), then the whole round against the round-5 tree
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for i in 1 2 3; do
  (cd _base && timeout 300 python tools/ab_prepare.py xl 1 --once base 2>&1 | grep -v "Warning\|amdgpu.ids" | sed 's/^/prev /')
  timeout 300 python tools/ab_prepare.py xl 1 --once base 2>&1 | grep -v "Warning\|amdgpu.ids" | sed 's/^/new  /'
done | tee gpurun_out/r06z_ab.txt
bash scripts/r06_ab_round.sh
