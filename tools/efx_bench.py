import torch, time, sys
sys.path.insert(0,'.')
from torchfx_amd import torchfx_ext as E
x=torch.randn(64,28_800_000,device="cuda:0")
def t(fn,n=10):
    fn(); torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): y=fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/n*1e3
b=8*x.numel()/1e9
for name,fn in (("gain",lambda:E.gain_forward(x,0.5,True)),("torch.mul",lambda:x*0.5),("normalize",lambda:E.normalize_forward(x,1.0,0,False)),("sum3",lambda:E.sum_forward([x,x,x]))):
    ms=t(fn); print(f"{name}: {ms:.3f} ms, {b/ms:.2f} TB/s (8 B/sample)")
