#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5_b10; mkdir -p $O
timeout 900 python tools/sos_ols_bench.py 7 fused,staged,wavepath 2>&1 | tee $O/sustained.txt
