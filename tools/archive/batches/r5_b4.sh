#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5_b4; mkdir -p $O
timeout 900 python tools/sos_ols_sweep2.py 2>&1 | tee $O/sweep.txt
