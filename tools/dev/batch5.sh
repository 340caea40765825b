mkdir -p gpurun_out/r2b5
python -m pytest tests/test_gpu_parity.py tests/test_gpu_reference_grids.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -3 > gpurun_out/r2b5/pytest_nf.txt
python tools/dev/iir_ceiling.py > gpurun_out/r2b5/iir_ceiling3.txt 2>&1
python bench.py --workload sos --no-extras > gpurun_out/r2b5/bench_sos.txt 2>&1
