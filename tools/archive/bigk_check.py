import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from torchfx_amd import torchfx_ext as E
from oracle import oracle as O
rng = np.random.default_rng(0)
for K in (33, 96, 97, 512, 513):
    r = rng.uniform(0.2, 0.95, K); th = rng.uniform(0.1, 3.0, K)
    sos = np.zeros((K, 6)); sos[:, 0] = rng.uniform(0.5, 1.0, K); sos[:, 1] = rng.uniform(-0.3, 0.3, K); sos[:, 2] = rng.uniform(-0.2, 0.2, K)
    sos[:, 3] = 1; sos[:, 4] = -2 * r * np.cos(th); sos[:, 5] = r * r
    # keep overall gain tame
    x = (rng.standard_normal((3, 50000)) * 1e-3).astype(np.float32)
    try:
        y, sx, sy = E.sos_forward(torch.from_numpy(x).cuda(), None, torch.from_numpy(sos), None, None, out_dtype=torch.float64)
        ey, esx, esy = O.sos_forward(x.astype(np.float64), sos)
        sc = max(1.0, np.abs(ey).max())
        print(K, "max rel err", np.abs(y.cpu().numpy() - ey).max() / sc, "scale", sc, "state err", np.abs(sy.cpu().numpy() - esy).max() / max(1, np.abs(esy).max()))
    except Exception as e:
        print(K, "ERR", repr(e)[:300])
