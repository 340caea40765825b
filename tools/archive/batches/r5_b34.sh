#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
( time timeout 3300 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 ) 2>&1
timeout 900 python tools/n20_short_rows.py 2>&1 | grep -v amdgpu | head -3
