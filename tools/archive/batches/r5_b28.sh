#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5_b28; mkdir -p $O
TFX_OLS_SOS_PIPE=1 timeout 900 python -m pytest tests/test_gpu_sos_ols.py -x -q 2>&1 | tail -4 | tee $O/pytest.txt
echo "== baseline (pipe off)"; timeout 600 python tools/sos_ols_bench.py 7 check,fused 2>&1 | grep "pass A\|max"
export TFX_OLS_SOS_PIPE=1
for cfg in "160 4" "160 2" "160 8" "160 16" "120 4" "80 4" "240 4" "96 4" "160 32"; do
  set -- $cfg
  echo "== pipe slab=$1 sub=$2"; TFX_OLS_SOS_PAIRS=$1 TFX_OLS_SOS_PIPE_PAIRS=$2 timeout 600 python tools/sos_ols_bench.py 7 check,fused 2>&1 | grep "pass A\|max"
done | tee $O/pipe.txt
