#!/bin/bash
# round 3, GPU batch 2: persistent prefetching row pass; why .ys ran at the IIR-kernel rate in batch 1
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
{
for v in "TFX_OLS_ROW_XCH=2" "TFX_OLS_ROW_XCH=1,TFX_OLS_ROW_PERSIST=4" "TFX_OLS_ROW_XCH=2,TFX_OLS_ROW_PERSIST=4"; do
  echo "== parity $v"; env $(echo $v | tr ',' ' ') timeout 600 python -m pytest tests/test_gpu_ols.py -m gpu -x -q -k "65536 or fftconv" 2>&1 | tail -2; done
timeout 900 python tools/ols_knobs.py "TFX_OLS_ROW_XCH=0" "TFX_OLS_ROW_XCH=1" "TFX_OLS_ROW_XCH=2" \
   "TFX_OLS_ROW_XCH=1,TFX_OLS_ROW_PERSIST=4" "TFX_OLS_ROW_XCH=1,TFX_OLS_ROW_PERSIST=8" "TFX_OLS_ROW_XCH=1,TFX_OLS_ROW_PERSIST=2" \
   "TFX_OLS_ROW_XCH=1,TFX_OLS_ROW_PERSIST=16" "TFX_OLS_ROW_XCH=2,TFX_OLS_ROW_PERSIST=4" "TFX_OLS_ROW_XCH=0,TFX_OLS_ROW_PERSIST=4" "TFX_OLS_ROW_XCH=1" 2>&1 | tail -12
python - <<'PY'
import time, torch, bench
from torchfx_amd import Wave
FS=48000
x=torch.randn(64,600*FS,device="cuda:0"); x/=x.abs().max()
f1,f2,fir,rev=bench.build_filters()
w=(Wave(x,FS,device=x.device)|f1|f2|fir|rev)
print("flags",w.fuse_fir,w.fuse_spectral,w.fuse_epilogue,[type(m).__name__+str(getattr(m,'kernel',torch.zeros(0)).numel()) for m in w.plan()])
plan,names=bench.plan_chain(x); print(names)
def timed(fn,n=10):
    out=fn(); torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(n): out=None; out=fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t)/n*1e3
print("run_plan", timed(lambda: bench.run_plan(plan,x)))
print("ys", timed(lambda: (Wave(x,FS,device=x.device)|f1|f2|fir|rev).ys))
print("run_plan", timed(lambda: bench.run_plan(plan,x)))
PY
} > gpurun_out/r3_b2.log 2>&1
tail -40 gpurun_out/r3_b2.log
