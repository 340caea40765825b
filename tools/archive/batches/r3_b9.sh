#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_gpu_streaming.py tests/test_gpu_multiprocess.py -m gpu -x -q 2>&1 | tail -15
timeout 900 python tools/stream_bench.py 2>&1 | tail -20
} > gpurun_out/r3_b9.log 2>&1
tail -50 gpurun_out/r3_b9.log
