mkdir -p gpurun_out/r2b5
python -m pytest tests/test_gpu_parity.py tests/test_gpu_reference_grids.py -x -q -m gpu -k "fft or chain or ols or fir or stream or epilogue" 2>&1 | tail -3 > gpurun_out/r2b5/pytest_tw.txt
python tools/ols_knobs.py "" "TFX_OLS_TWCOL=0" "" "TFX_OLS_TWCOL=0" > gpurun_out/r2b5/knobs_tw.txt 2>&1
