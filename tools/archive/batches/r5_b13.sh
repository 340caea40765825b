#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5_b13; mkdir -p $O
timeout 900 python tools/sos_ols_sweep3.py 2>&1 | tee $O/sweep3.txt
