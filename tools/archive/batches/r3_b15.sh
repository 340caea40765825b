#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
{
TFX_OLS_FUSED=1 timeout 900 python -m pytest tests/test_gpu_ols.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -3
args=("TFX_OLS_FUSED=0")
for fs in 2 3; do for p in 2 3 4 5 6 8; do args+=("TFX_OLS_FUSED=1,TFX_OLS_FUSED_STREAMS=$fs,TFX_OLS_FUSED_PAIRS=$p"); done; done
python tools/ols_wall.py "${args[@]}" "TFX_OLS_FUSED=0" 2>&1 | grep -v amdgpu
} > gpurun_out/r3_b15.log 2>&1
cat gpurun_out/r3_b15.log
