mkdir -p gpurun_out/r2b10
python - > gpurun_out/r2b10/sos_ab.txt 2>&1 <<'PY'
import os, sys, torch
sys.path.insert(0, ".")
from tools.quick_bench import timed
from torchfx_amd import torchfx_ext as E
import bench
f1, f2, _, _ = bench.build_filters()
sos = torch.cat([f1._sos, f2._sos])
for shape in ((64, 2_880_000), (64, 28_800_000), (8, 2_880_000), (512, 480_000)):
    x = torch.randn(*shape, device="cuda")
    for rep in range(3):
        out = []
        for var in (2, 4, 0):
            os.environ["TFX_SOS_VARIANT"] = str(var)
            wall, prof = timed(lambda: E.sos_forward(x, None, sos, None, None), reps=10, warm=3)
            out.append(f"v{var} {list(prof.values())[0]:.4f}")
        print(shape, " | ".join(out), flush=True)
    del x
PY
