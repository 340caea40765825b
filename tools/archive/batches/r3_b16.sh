#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
{
TFX_OLS_ROW_XCH=3 timeout 900 python -m pytest tests/test_gpu_ols.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -3
python tools/ols_wall.py "TFX_OLS_ROW_XCH=2" "TFX_OLS_ROW_XCH=3" "TFX_OLS_ROW_XCH=2" "TFX_OLS_ROW_XCH=3" "TFX_OLS_ROW_XCH=3,TFX_OLS_STREAMS=3,TFX_OLS_SLAB_MB=48" "TFX_OLS_ROW_XCH=3,TFX_OLS_PAIRS_PER_SLAB=6" 2>&1 | grep -v amdgpu
python tools/ols_knobs.py "TFX_OLS_ROW_XCH=2,TFX_OLS_SLAB_MB=1024,TFX_OLS_STREAMS=3" "TFX_OLS_ROW_XCH=3,TFX_OLS_SLAB_MB=1024,TFX_OLS_STREAMS=3" 2>&1 | grep -v amdgpu
} > gpurun_out/r3_b16.log 2>&1
cat gpurun_out/r3_b16.log
