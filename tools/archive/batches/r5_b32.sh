#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5_b32; mkdir -p $O
TFX_OLS_SOS_PHASED=1 timeout 900 python -m pytest tests/test_gpu_sos_ols.py -x -q 2>&1 | tail -3
echo "== baseline"; timeout 600 python tools/sos_ols_bench.py 7 check,fused,sustained 2>&1 | grep "pass A\|max" | head -4
export TFX_OLS_SOS_PHASED=1
for sub in 4 8 2 16; do
  echo "== phased, pieces of $sub pairs"; TFX_OLS_SOS_PIPE_PAIRS=$sub timeout 600 python tools/sos_ols_bench.py 7 check,fused,sustained 2>&1 | grep "pass A\|max" | head -4
done | tee $O/phased.txt
