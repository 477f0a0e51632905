#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for sh in geglu geglu_b4; do for c in "pp 128x288" "pp60 LN" "co"; do timeout 300 tools/_run/gemm_bench $sh "$c"; done; done 2>&1 | grep -v "^==" | cut -c1-330 | tee gpurun_out/r06b_gemm_bench.txt
