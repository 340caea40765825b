#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5_final; mkdir -p $O
( time timeout 3300 python -m pytest tests -m gpu -q 2>&1 | tail -8 ) 2>&1 | tee $O/pytest.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $O/smoke.txt
bash tools/batches/r5_profiles.sh > $O/profiles.log 2>&1
tail -5 $O/profiles.log
