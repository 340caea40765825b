#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5_b16; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_sos_ols.py tests/test_gpu_ols.py -x -q 2>&1 | tail -6 | tee $O/pytest.txt
timeout 1500 python tools/n21_sweep.py 2>&1 | grep -v amdgpu | tee $O/sweep.txt
