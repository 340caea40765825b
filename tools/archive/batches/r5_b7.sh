#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5_b7; mkdir -p $O
for prio in 0 1 0 1; do
  echo "== prio $prio"; TFX_OLS_SOS_PRIO=$prio timeout 300 python tools/sos_ols_bench.py 7 fused 2>&1 | grep -v amdgpu
done | tee $O/prio.txt
echo "== single lane, 512 pairs"
for prio in 0 1; do TFX_OLS_SOS_PRIO=$prio TFX_OLS_SOS_STREAMS=1 TFX_OLS_SOS_PAIRS=512 timeout 300 python tools/sos_ols_bench.py 5 fused 2>&1 | grep -v amdgpu; done | tee -a $O/prio.txt
