#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
{
/opt/rocm/bin/rocm-smi --showclocks --showpower 2>&1 | grep -E "sclk|mclk|Power" | head -6
for rep in 1 2 3; do for v in 0 2; do
  echo -n "XCH=$v rep=$rep: "; TFX_OLS_ROW_XCH=$v python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(l['ms_per_step'], {k:v['ms_per_step'] for k,v in l['kernels'].items()})"
done; done
} > gpurun_out/r3_b7.log 2>&1
cat gpurun_out/r3_b7.log
