#!/usr/bin/env python
"""Wall-time of the spin-temperature filter stage on one MI355X (device-resident):
fill_Rbox_table over N_STEP_TS radii and one UpdateXraySourceBox shell (two grids), straight-
line and multiple-scattering windows.  Algorithmic bytes as in SURVEY 8(d): per radius one
filtered grid = 5 S (copy x filter, one ideal c2r pass, pass-Z read) + 4N stored."""
import importlib, json, sys, time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch

api = importlib.import_module("21cmfast_amd.grid_api")
S = importlib.import_module("21cmfast_amd.structs")

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
n_R = int(sys.argv[2]) if len(sys.argv) > 2 else 40
L = 1.5 * n
g = torch.Generator(device="cuda").manual_seed(1)
f = torch.rand((n, n, n), device="cuda", generator=g) + 0.1
f2 = torch.rand((n, n, n), device="cuda", generator=g) + 0.1
radii = list(0.62 * 1.5 * 1.16 ** np.arange(n_R))


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


N = float(n) ** 3
res = {"hii_dim": n, "n_R": n_R}
spec = S.rbox_spec(n, L, radii, filter_type=0, min_value=-1.0, const_factor=1.0)
res["fill_Rbox_ms"] = timed(lambda: api.fill_Rbox_grids(spec, f))
res["fill_Rbox_ms_per_radius"] = res["fill_Rbox_ms"] / n_R
res["fill_Rbox_alg_TBps"] = (20 + 4) * N * n_R / (res["fill_Rbox_ms"] * 1e-3) / 1e12
for name, types, rs in (("shell_sl", [4, 4], 0.0), ("shell_ms", [5, 4], 8.0),
                        ("shell_ms_mini", [5, 4, 5, 4, 4], 8.0)):
    sp = S.annular_spec(n, L, 20.0, 24.0, types, R_star=rs)
    grids = [f, f2, f, f2, f][: len(types)]
    res[name + "_ms"] = timed(lambda: api.annular_filter_grids(sp, grids))
print(json.dumps(res))
