#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
{
for v in 1 2 1 4 1; do
  echo -n "TFX_FIR_NJ=$v: "; TFX_FIR_NJ=$v python bench.py --workload fir --steps 100 --warmup 5 --no-cpu-baseline --no-extras | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step', l['ms_per_step'], 'frac', l['roofline']['frac'])"
done
TFX_FIR_NJ=1 timeout 600 python -m pytest tests/test_gpu_fir.py tests/test_gpu_fullsize.py -m gpu -x -q -k "fir" 2>&1 | tail -2
} > gpurun_out/r3_b12.log 2>&1
cat gpurun_out/r3_b12.log | grep -v amdgpu.ids
