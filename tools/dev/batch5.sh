mkdir -p gpurun_out/r2b5
python -m pytest tests/test_gpu_parity.py tests/test_gpu_reference_grids.py tests/test_gpu_reference_grids2.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -3 > gpurun_out/r2b5/pytest_cache.txt
python tools/dev/stall_probe5.py > gpurun_out/r2b5/stall5.txt 2>&1
