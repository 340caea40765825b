#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5_b19; mkdir -p $O
for n21 in 1 0 1 0; do
TFX_OLS_SOS_N21=$n21 timeout 900 python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench chain N21=$n21', d['ms_per_step'], d['roofline']['frac'], d['config']['overlap_save'], {k:(v['ms_per_step'], v.get('GBps')) for k,v in d['kernels'].items()})"
done | tee $O/bench_n21.txt
