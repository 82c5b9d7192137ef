#!/usr/bin/env python
"""Wall-time of one ComputeIonizedBox grid pass with an x_e grid (USE_TS_FLUCT: three filtered
grids per radius) next to the default two-grid pass, device-resident, 40 radii."""
import importlib, json, sys, time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

api = importlib.import_module("21cmfast_amd.grid_api")
W = importlib.import_module("21cmfast_amd.workloads")

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
density = W.density_field_torch(n, seed=12345)
n_ion = W.nion_from_density(density)
g = torch.Generator(device="cuda").manual_seed(3)
xe = 0.02 + 0.03 * torch.rand((n, n, n), device="cuda", generator=g)
Tn = 8.0 + 4.0 * torch.rand((n, n, n), device="cuda", generator=g)
res = {"hii_dim": n}
for name, ts in (("two_grids", 0), ("with_xe", 1)):
    spec = W.ionize_spec(n, use_ts_fluct=ts)
    buf = api.IonizeBuffers(density)
    kw = dict(xe=xe, Tneutral=Tn) if ts else {}

    def step():
        buf.reset()
        return api.ionize_grids(spec, density, n_ion, buffers=buf, **kw)[2]

    step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        rep = step()
    torch.cuda.synchronize()
    res[name + "_ms"] = (time.perf_counter() - t0) / 3 * 1e3
    res[name + "_xH"] = rep.global_xH
print(json.dumps(res))
