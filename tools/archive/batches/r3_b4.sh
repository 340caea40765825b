#!/bin/bash
# round 3, GPU batch 4: the whole -m gpu suite after the sos non-finite flags, LC64 taps, gather rewrite, new tests
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 2300 python -m pytest tests/ -m gpu -x -q --durations=15 > gpurun_out/r3_b4.log 2>&1
tail -40 gpurun_out/r3_b4.log
