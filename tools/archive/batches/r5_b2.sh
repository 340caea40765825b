#!/bin/bash
# round 5, batch 2: the 128-register build of the cascade-in-pass-A kernel (two workgroups per CU): parity, sweep, kernel trace
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5_b2; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_sos_ols.py -x -q 2>&1 | tail -5 | tee $O/pytest.txt
timeout 900 python tools/sos_ols_bench.py 5 check,fused,staged,ols,sweep 2>&1 | tee $O/bench.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/trace -o t -- python $GRAFT_REPO_ROOT/tools/sos_ols_bench.py 3 fused > $GRAFT_REPO_ROOT/$O/trace.log 2>&1
cd $GRAFT_REPO_ROOT; ls -R $O/trace | head; python tools/trace_timeline.py $O/trace 2>&1 | tee $O/timeline.txt | tail -40
