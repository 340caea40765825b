#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5_b8; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_sos_ols.py -x -q 2>&1 | tail -3
for sub in 0 8 16 32 0 8; do
  echo "== sub-slab pairs $sub"; TFX_OLS_SOS_SUB_PAIRS=$sub timeout 300 python tools/sos_ols_bench.py 7 fused 2>&1 | grep -v amdgpu
done | tee $O/sub.txt
