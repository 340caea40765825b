#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 2400 python -m pytest tests/test_gpu_sos_ols.py -x -q -k random_geometry 2>&1 | tail -12
