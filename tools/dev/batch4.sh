mkdir -p gpurun_out/r2b4
python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r2b4/pytest_gpu.txt
python bench.py > gpurun_out/r2b4/bench_chain.json 2> gpurun_out/r2b4/bench_chain.err
