#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5_b11; mkdir -p $O
for i in 1 2 3; do
timeout 900 python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench chain', d['ms_per_step'], d['ms_per_step_with_event_profiling'])"
done | tee $O/bench3.txt
timeout 900 python tools/sos_ols_bench.py 7 wavepath 2>&1 | grep -v amdgpu | head -6 | tee -a $O/bench3.txt
timeout 900 python bench.py --steps 20 --warmup 3 --workload chain_iir_kernel --no-extras --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench chain_iir_kernel', d['ms_per_step'])" | tee -a $O/bench3.txt
