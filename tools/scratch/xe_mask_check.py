"""Eulerian table model with an x_e grid (config 5's IonizedBox): fused x_e pass Z + mask against the
dense-x_e path (C21CM_XE_MASK_FUSED=0): identical outputs; timings."""
import importlib, os, sys, time
from pathlib import Path
root = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(root)); sys.path.insert(0, str(root / "tests"))
import numpy as np, torch
W = importlib.import_module("21cmfast_amd.workloads")
S = importlib.import_module("21cmfast_amd.structs")
api = importlib.import_module("21cmfast_amd.grid_api")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
spec = W.ionize_spec(n, mode=W.FCOLL_TABLE_EXP if hasattr(W, "FCOLL_TABLE_EXP") else 3, r_bubble_max=30.0)
spec.use_ts_fluct = 1
spec.hii_filter = 0
def table_fn(r_index, dmin, dmax, table, user):
    x = np.linspace(dmin, dmax, S.NDELTA_TABLE)
    y = np.log(0.02 * (1 + np.clip(x, -0.999, None)) ** 1.5 / (1 + 0.05 * r_index) + 1e-30)
    for i in range(S.NDELTA_TABLE): table[i] = y[i]
    return 0
cb = S.TABLE_FN(table_fn); spec.table_fn = cb
density = W.density_field_torch(n)
g = torch.Generator(device="cuda").manual_seed(5)
xe = (0.3 * torch.rand(density.shape, device="cuda", generator=g)).float()
Tn = (50 + 10 * torch.rand(density.shape, device="cuda", generator=g)).float()
out = {}
for mode in ("0", "1"):
    os.environ["C21CM_XE_MASK_FUSED"] = mode
    ts = []
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        buf, box, r = api.ionize_grids(spec, density, None, xe=xe, Tneutral=Tn)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    out[mode] = (buf.neutral_fraction.clone(), buf.z_reion.clone(), r.global_xH, min(ts))
    print("fused" if mode == "1" else "dense", "ms", round(min(ts), 2), "xH", r.global_xH, "ionised", float((buf.neutral_fraction == 0).float().mean()))
print("identical x_HI:", bool(torch.equal(out["0"][0], out["1"][0])), " z_reion:", bool(torch.equal(out["0"][1], out["1"][1])))
