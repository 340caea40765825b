set -x
mkdir -p gpurun_out/r2b1
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fft or chain or ols or fir" 2>&1 | tail -5 > gpurun_out/r2b1/pytest_default.txt
TFX_OLS_COL_THREADS=512 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fft or chain or ols or fir" 2>&1 | tail -5 > gpurun_out/r2b1/pytest_t512.txt
TFX_OLS_ROWMAP=0 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fft or chain or ols" 2>&1 | tail -5 > gpurun_out/r2b1/pytest_map0.txt
./tools/ubench/bin/seg_pattern > gpurun_out/r2b1/seg_pattern.txt 2>&1
python tools/ols_knobs.py "TFX_OLS_ROWMAP=0" "" "TFX_OLS_COL_THREADS=512" "TFX_OLS_SLAB_MB=128" "TFX_OLS_SLAB_MB=256" "TFX_OLS_SLAB_MB=32" "TFX_OLS_SLAB_MB=128,TFX_OLS_COL_THREADS=512" "TFX_OLS_STREAMS=3" "TFX_OLS_STREAMS=3,TFX_OLS_COL_THREADS=512" "TFX_OLS_STREAMS=4,TFX_OLS_SLAB_MB=32" "TFX_OLS_PROBE=1" "TFX_OLS_PROBE=2" "TFX_OLS_PROBE=3" "TFX_OLS_PROBE=1,TFX_OLS_COL_THREADS=512" "TFX_OLS_PROBE=2,TFX_OLS_COL_THREADS=512" "TFX_OLS_PROBE=3,TFX_OLS_COL_THREADS=512" > gpurun_out/r2b1/knobs.txt 2>&1
python bench.py --no-cpu-baseline > gpurun_out/r2b1/bench_chain.json 2> gpurun_out/r2b1/bench_chain.err
