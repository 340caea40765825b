#!/bin/bash
# round 5, batch 1: parity of the cascade-in-pass-A kernel, then its first timings
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r5_b1
timeout 900 python -m pytest tests/test_gpu_sos_ols.py -x -q 2>&1 | tail -25 > gpurun_out/r5_b1/pytest.txt
cat gpurun_out/r5_b1/pytest.txt
timeout 600 python tools/sos_ols_bench.py 5 check,fused,staged,ols,sweep 2>&1 | tee gpurun_out/r5_b1/bench.txt
