#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5_b6; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_sos_ols.py -x -q 2>&1 | tail -5 | tee $O/pytest.txt
timeout 900 python tools/sos_ols_bench.py 7 check,fused,staged 2>&1 | tee $O/bench.txt
for bits in 60 48 40; do for unit in 0 1; do
  echo "== halo bits $bits unit $unit"; TFX_OLS_SOS_HALO_BITS=$bits TFX_OLS_SOS_UNIT_B0=$unit timeout 300 python tools/sos_ols_bench.py 7 fused 2>&1 | grep -v amdgpu
done; done | tee $O/knobs.txt
