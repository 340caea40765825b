#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5_b25; mkdir -p $O
for cfg in "3 0" "3 240" "2 240" "3 160" "4 120" "3 0" "3 240" "2 240" "3 480"; do
  set -- $cfg
  echo "== streams=$1 pairs=$2 (0 = default policy)"
  if [ "$2" = "0" ]; then TFX_OLS_SOS_STREAMS=$1 timeout 600 python tools/sos_ols_bench.py 7 fused,sustained 2>&1 | grep "pass A" | head -3
  else TFX_OLS_SOS_STREAMS=$1 TFX_OLS_SOS_PAIRS=$2 timeout 600 python tools/sos_ols_bench.py 7 fused,sustained 2>&1 | grep "pass A" | head -3; fi
done | tee $O/slabs.txt
