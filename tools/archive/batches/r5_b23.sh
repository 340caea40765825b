#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5_b23; mkdir -p $O
TFX_OLS_TRACE=1 timeout 600 python bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline > $O/out.txt 2> $O/err.txt
grep "tfx ols" $O/err.txt | head -12; python -c "
import json; d=json.loads(open('$O/out.txt').read().strip().splitlines()[-1]); print(d['end_to_end'])"
TFX_OLS_TRACE=1 TORCHFX_AMD_FUSE_RECURSIVE=0 timeout 600 python bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline > $O/out2.txt 2> $O/err2.txt
grep "tfx ols" $O/err2.txt | head -12; python -c "
import json; d=json.loads(open('$O/out2.txt').read().strip().splitlines()[-1]); print(d['end_to_end'])"
