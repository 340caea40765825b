#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5_b14; mkdir -p $O
( time timeout 3300 python -m pytest tests -m gpu -q 2>&1 | tail -40 ) 2>&1 | tee $O/pytest.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee $O/smoke.txt
