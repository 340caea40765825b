#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 900 python -m pytest tests/test_gpu_multiprocess.py -x -q -k plain_c_host 2>&1 | tail -8
