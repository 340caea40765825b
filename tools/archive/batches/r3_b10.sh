#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 2300 python -m pytest tests/ -m gpu -x -q > gpurun_out/r3_b10.log 2>&1
tail -12 gpurun_out/r3_b10.log
