#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5_b17; mkdir -p $O
timeout 900 python tools/n21_probe.py 2>&1 | grep -v amdgpu | tee $O/probe.txt
