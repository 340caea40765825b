#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5_b21; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_sos_ols.py tests/test_gpu_ols.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -6 | tee $O/pytest.txt
for i in 1 2; do
timeout 900 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench chain', d['ms_per_step'], d['roofline']['frac'], d['end_to_end'])"
done | tee $O/bench.txt
TFX_OLS_TRACE=1 timeout 300 python tools/first_call.py 2>&1 | tail -25 | tee $O/first_call.txt
