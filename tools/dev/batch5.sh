mkdir -p gpurun_out/r2b5
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "cfg1 or cfg2 or golden or segmentation or random_shapes or bank or epilogue or non_finite" 2>&1 | tail -3 > gpurun_out/r2b5/pytest_pf.txt
python tools/dev/iir_ceiling.py > gpurun_out/r2b5/iir_ceiling_pf.txt 2>&1
