mkdir -p gpurun_out/r2b3
./tools/ubench/bin/mall2 > gpurun_out/r2b3/mall2.txt 2>&1
python tools/quick_bench.py sos > gpurun_out/r2b3/sos.txt 2>&1
