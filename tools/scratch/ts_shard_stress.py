"""Repeat single pass vs emulated shard phases (Eulerian table model + x_e grid) on FIXED inputs and
count mismatching cells per repeat -- any non-zero count is a race, not arithmetic (diagnostic)."""
import importlib, sys, pathlib, os
import numpy as np, torch
sys.path.insert(0, str(pathlib.Path.cwd()))
pkg = importlib.import_module("21cmfast_amd")
W = importlib.import_module("21cmfast_amd.workloads")
api = importlib.import_module("21cmfast_amd.grid_api")
S = importlib.import_module("21cmfast_amd.structs")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
def table_fn(r_index, dmin, dmax, table, user):
    x = dmin + (dmax - dmin) / (S.NDELTA_TABLE - 1.0) * np.arange(S.NDELTA_TABLE)
    y = np.log(0.02 * (1 + np.maximum(x, -0.999)) ** 1.5 / (1 + 0.05 * r_index))
    for i in range(S.NDELTA_TABLE):
        table[i] = y[i]
    return 0
cb = S.TABLE_FN(table_fn)
spec = W.ionize_spec(n, mode=W.FCOLL_TABLE_EXP, r_bubble_max=30.0, use_ts_fluct=1)
spec.hii_filter = 0
spec.table_fn = cb
rng = np.random.default_rng(11)
density = W.density_field_numpy(n, seed=7)
xe = (0.3 * rng.random((n, n, n))).astype(np.float32)
Tn = (50 + 10 * rng.random((n, n, n))).astype(np.float32)
d, x, t = (torch.from_numpy(a).cuda() for a in (density, xe, Tn))
ref = None
bad = 0
for rep in range(reps):
    buf0, _, rep0 = api.ionize_grids(spec, d, None, xe=x, Tneutral=t)
    torch.cuda.synchronize()
    a0 = buf0.neutral_fraction.clone()
    if ref is None:
        ref = a0.clone()
    ns = int((a0 != ref).sum())
    world = 2
    masks = []
    for rank in range(world):
        fc = torch.zeros((n, n, n), dtype=torch.uint8, device="cuda")
        api.ionize_shard_radii(spec, rank, world, fc, d, None, xe=x, Tneutral=t, want_report=False)
        masks.append(fc)
    red = torch.maximum(masks[0], masks[1]).contiguous()
    buf2, _, rep2 = api.ionize_shard_finish(spec, red, d, None, xe=x, Tneutral=t)
    torch.cuda.synchronize()
    nd = int((ref != buf2.neutral_fraction).sum())
    if ns or nd:
        bad += 1
        print("rep", rep, "single vs first single:", ns, " sharded vs first single:", nd, flush=True)
print("repeats", reps, "with a mismatch:", bad, "ionised fraction", float((ref == 0).float().mean()))
