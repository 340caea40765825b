#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_gpu_ols.py tests/test_gpu_fullsize.py tests/test_gpu_effects.py -m gpu -x -q 2>&1 | tail -4
python tools/ols_wall.py "TFX_OLS_FUSED=0" "TFX_OLS_FUSED_PAIRS=2" "TFX_OLS_FUSED_PAIRS=3" "TFX_OLS_FUSED_PAIRS=4" "TFX_OLS_FUSED_PAIRS=5" "TFX_OLS_FUSED_PAIRS=6" "TFX_OLS_FUSED_PAIRS=8" "TFX_OLS_FUSED_PAIRS=12" "TFX_OLS_FUSED_PAIRS=32" "TFX_OLS_FUSED=0" 2>&1 | grep -v amdgpu
} > gpurun_out/r3_b14.log 2>&1
cat gpurun_out/r3_b14.log
