#!/bin/bash
# round 3, GPU batch 1: pass-B exchange layouts (parity + timing), plan-cache .ys rate, seam windows
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_gpu_ols.py -m gpu -x -q -k "fft or ols or conv or chain" 2>&1 | tail -5
for v in 0 1; do echo "== XCH=$v"; TFX_OLS_ROW_XCH=$v timeout 600 python -m pytest tests/test_gpu_ols.py -m gpu -x -q -k "65536 or fftconv" 2>&1 | tail -2; done
timeout 900 python tools/ols_knobs.py "TFX_OLS_ROW_XCH=0" "TFX_OLS_ROW_XCH=1" "TFX_OLS_ROW_XCH=2" "TFX_OLS_ROW_XCH=0" "TFX_OLS_ROW_XCH=2" 2>&1 | tail -8
timeout 1200 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -s -k "cfg4 or cfg5_chain or wave_ys" 2>&1 | tail -12
} > gpurun_out/r3_b1.log 2>&1
tail -40 gpurun_out/r3_b1.log
