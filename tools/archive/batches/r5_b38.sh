#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 2400 python tools/soak.py 40 2>&1 | grep -v amdgpu | tee gpurun_out/r5_soak.txt | tail -45
