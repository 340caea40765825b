#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5_b24; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_sos_ols.py tests/test_gpu_ols.py -x -q 2>&1 | tail -4
for rm in 1 0 1 0; do
  echo "== TFX_OLS_ROWMAP=$rm"; TFX_OLS_ROWMAP=$rm timeout 600 python tools/sos_ols_bench.py 7 fused,sustained 2>&1 | grep "pass A"
done | tee $O/rowmap.txt
