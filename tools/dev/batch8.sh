mkdir -p gpurun_out/r2b8
python tools/stream_bench.py > gpurun_out/r2b8/stream_bench.txt 2>&1
python bench.py > gpurun_out/r2b8/bench_chain.json 2> gpurun_out/r2b8/bench_chain.err
