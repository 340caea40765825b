#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_gpu_iir.py tests/test_gpu_bank_sum.py tests/test_gpu_effects.py -m gpu -x -q 2>&1 | tail -4
python tools/sos_knobs.py "" "" 2>&1 | grep -v amdgpu
python - <<'PY'
import time, torch, bench
from torchfx_amd import torchfx_ext as E
f1,f2,_,_=bench.build_filters(); sos=torch.cat([f1._sos,f2._sos]).contiguous()
x=torch.randn(64,2_880_000,device="cuda:0")
for prec in ("f64","f32"):
    for _ in range(5): E.sos_forward(x,None,sos,None,None,precision=prec)
    torch.cuda.synchronize(); ts=[]
    for rep in range(5):
        t0=time.perf_counter()
        for _ in range(20): y=E.sos_forward(x,None,sos,None,None,precision=prec)
        torch.cuda.synchronize(); ts.append((time.perf_counter()-t0)/20*1e3)
    print(prec,"wall ms per step, 20 back-to-back:", [round(t,4) for t in ts])
PY
} > gpurun_out/r3_b13.log 2>&1
cat gpurun_out/r3_b13.log
