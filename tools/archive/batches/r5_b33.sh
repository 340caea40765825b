#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 900 python tools/n20_short_rows.py 2>&1 | grep -v amdgpu | tee gpurun_out/r5_b33.txt
