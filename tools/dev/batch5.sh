mkdir -p gpurun_out/r2b5
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fft or chain or ols or fir" 2>&1 | tail -3 > gpurun_out/r2b5/pytest_pk.txt
python tools/ols_knobs.py "" "TFX_OLS_ROW_HPRE=1" "TFX_OLS_COL_THREADS=256" "" > gpurun_out/r2b5/knobs4.txt 2>&1
