#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5_b15; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_sos_ols.py tests/test_gpu_ols.py -x -q 2>&1 | tail -15 | tee $O/pytest.txt
for n21 in 0 1; do
  echo "== TFX_OLS_N21=$n21"
  TFX_OLS_N21=$n21 timeout 600 python tools/sos_ols_bench.py 7 check,fused,staged,ols 2>&1 | grep -v amdgpu
done | tee $O/n21.txt
