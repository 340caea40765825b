#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5_b18; mkdir -p $O
for n21 in 1 0 1 0; do
  echo "== TFX_OLS_SOS_N21=$n21"
  TFX_OLS_SOS_N21=$n21 timeout 600 python tools/sos_ols_bench.py 7 check,fused,sustained 2>&1 | grep -v amdgpu | grep "check\|max\|pass A"
done | tee $O/n21.txt
timeout 900 python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench chain', d['ms_per_step'], d['roofline']['frac'], d['config']['overlap_save'])" | tee -a $O/n21.txt
