#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5_b22; mkdir -p $O
TFX_OLS_TRACE=1 timeout 300 python tools/first_call.py > $O/out.txt 2> $O/err.txt
grep "tfx ols" $O/err.txt | head -40; grep "first\|second" $O/out.txt | head
