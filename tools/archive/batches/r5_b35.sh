#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5_b35; mkdir -p $O
for i in 1 2 3; do
TFX_OLS_TRACE=1 timeout 600 python bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline > $O/out$i.txt 2> $O/err$i.txt
python -c "
import json; d=json.loads(open('$O/out$i.txt').read().strip().splitlines()[-1]); print('run $i first_ys_ms', d['end_to_end']['first_ys_ms'])"
grep "tfx ols" $O/err$i.txt | head -9 | awk '{ if (\$NF+0 > 1.0 || 1) print }' | tr '\n' ';'; echo
done
