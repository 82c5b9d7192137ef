import importlib, os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from recomb_helpers import inputs, recomb_spec
api = importlib.import_module("21cmfast_amd.grid_api")
n = 256
spec = recomb_spec(n, model=2, cell_recomb=1, r_bubble_max=20.0)
d = {k: torch.from_numpy(v).cuda() for k, v in inputs((n, n, n), seed=77).items()}
kw = dict(n_ion=d["n_ion"], whalo_sfr=d["whalo_sfr"], prev_nrec=d["prev_nrec"], prev_z_reion=d["prev_z_reion"])
os.environ["C21CM_RECOMB_FUSED"] = "0"
b0, _, r0 = api.ionize_grids(spec, d["density"], **kw)
m0 = b0.mean_free_path.clone(); g0 = b0.ionisation_rate_G12.clone()
del os.environ["C21CM_RECOMB_FUSED"]
for pair in ("1", "0"):
    os.environ["C21CM_PAIR_RADII"] = pair
    b1, _, r1 = api.ionize_grids(spec, d["density"], **kw)
    m1 = b1.mean_free_path; g1 = b1.ionisation_rate_G12
    diff = (m0 != m1)
    print("pair", pair, "cells with different mfp:", int(diff.sum()), "of", m0.numel())
    R = np.array(spec.R[:spec.n_radii], dtype=np.float32)
    if diff.any():
        a = m0[diff].cpu().numpy(); b = m1[diff].cpu().numpy()
        ia = np.searchsorted(R, a); ib = np.searchsorted(R, b)
        print(" unfused idx hist", np.bincount(ia, minlength=spec.n_radii))
        print(" fused   idx hist", np.bincount(ib, minlength=spec.n_radii))
    same = ~diff
    rel = ((g0[same] - g1[same]).abs() / g0[same].abs().clamp_min(1e-20))
    print(" G12 max rel diff where same mfp:", float(rel.max()))
