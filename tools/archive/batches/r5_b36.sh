#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for nt in 3 0 1 2 3 0; do
  echo "== TFX_OLS_NT=$nt"; TFX_OLS_NT=$nt timeout 600 python tools/sos_ols_bench.py 7 fused,sustained 2>&1 | grep "pass A" | head -3
done | tee gpurun_out/r5_b36.txt
