mkdir -p gpurun_out/r2b5
python -m pytest tests/test_gpu_parity.py tests/test_gpu_reference_grids.py -x -q -m gpu -k "fir or stream or realtime or process_file or Fir" 2>&1 | tail -3 > gpurun_out/r2b5/pytest_sf.txt
python tools/stream_bench.py > gpurun_out/r2b5/stream_bench3.txt 2>&1
