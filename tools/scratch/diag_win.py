import importlib, os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
api = importlib.import_module("21cmfast_amd.grid_api")
S = importlib.import_module("21cmfast_amd.structs")
n = 256
rng = np.random.default_rng(3)
f = torch.from_numpy((0.4 * rng.standard_normal((n, n, n))).astype(np.float32)).cuda()
radii = [0.5, 0.95, 1.3, 2.0, 3.7, 6.0, 11.0, 19.0, 33.0]
for ft in (0, 1):
    spec = S.rbox_spec(n, 1.5 * n, radii, filter_type=ft, min_value=-10.0, const_factor=1.0)
    os.environ["C21CM_WINDOWS"] = "table"
    tab = api.fill_Rbox_grids(spec, f)["result"].clone()
    del os.environ["C21CM_WINDOWS"]
    ev = api.fill_Rbox_grids(spec, f)["result"]
    for i, R in enumerate(radii):
        a, b = tab[i].double(), ev[i].double()
        rms = float(a.std())
        print(f"filter {ft} R={R:5.2f}: rms {rms:.3e} max|diff|/rms {float((a-b).abs().max())/rms:.2e}  rms diff/rms {float((a-b).std())/rms:.2e}")
