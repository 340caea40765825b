#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 1500 python -m pytest tests/test_gpu_planner.py -x -q -k "gain_and_normalize_ride or default_plan_runs or fold_error" 2>&1 | tail -8
