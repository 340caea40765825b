#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
{
for v in "TFX_OLS_ROW_XCH=1,TFX_OLS_ROW_HEARLY=1" "TFX_OLS_ROW_XCH=2,TFX_OLS_ROW_HEARLY=1"; do
  echo "== parity $v"; env $(echo $v | tr ',' ' ') timeout 600 python -m pytest tests/test_gpu_ols.py -m gpu -x -q -k "65536 or fftconv" 2>&1 | tail -2; done
timeout 900 python tools/ols_knobs.py "TFX_OLS_ROW_XCH=0" "TFX_OLS_ROW_XCH=2" "TFX_OLS_ROW_XCH=1,TFX_OLS_ROW_HEARLY=1" "TFX_OLS_ROW_XCH=2,TFX_OLS_ROW_HEARLY=1" "TFX_OLS_ROW_XCH=2" "TFX_OLS_ROW_XCH=1,TFX_OLS_ROW_HEARLY=1" "TFX_OLS_ROW_XCH=1"  2>&1 | tail -8
} > gpurun_out/r3_b3.log 2>&1
tail -40 gpurun_out/r3_b3.log
