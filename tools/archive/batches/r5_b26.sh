#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5_b26; mkdir -p $O
TFX_OLS_SOS_SPLIT=1 timeout 900 python -m pytest tests/test_gpu_sos_ols.py -x -q 2>&1 | tail -4 | tee $O/pytest.txt
for sp in 1 0 1 0; do
  echo "== TFX_OLS_SOS_SPLIT=$sp"; TFX_OLS_SOS_SPLIT=$sp timeout 600 python tools/sos_ols_bench.py 7 check,fused,sustained 2>&1 | grep "pass A\|max" | head -4
done | tee $O/split.txt
echo "== single lane"; for sp in 1 0; do TFX_OLS_SOS_SPLIT=$sp TFX_OLS_SOS_STREAMS=1 timeout 600 python tools/sos_ols_bench.py 5 fused 2>&1 | grep "pass A"; done | tee -a $O/split.txt
