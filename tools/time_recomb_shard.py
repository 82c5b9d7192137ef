"""The phases of the sharded fused recombination loop at HII_DIM^3 (40 radii, inhomogeneous
CELL_RECOMB), timed one after the other on one GPU for every rank of `world` (diagnostic, GPU box
only): shard phase per rank, the slab combine, the finish phase; and the key phases for comparison.
usage: python tools/time_recomb_shard.py [HII_DIM] [world]"""
import importlib
import json
import sys
import time
from pathlib import Path

root = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(root))
sys.path.insert(0, str(root / "tests"))
import torch  # noqa: E402

import recomb_helpers as RH  # noqa: E402

W = importlib.import_module("21cmfast_amd.workloads")
D = importlib.import_module("21cmfast_amd.distributed")
api = importlib.import_module("21cmfast_amd.grid_api")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
density = W.density_field_torch(n)
n_ion = W.nion_from_density(density)
g = torch.Generator(device="cuda").manual_seed(3)
whalo = (n_ion * (0.8 + 0.4 * torch.rand(density.shape, device="cuda", generator=g)) * 1e-9).float()
prev_nrec = (0.6 * torch.rand(density.shape, device="cuda", generator=g) ** 2).float()
prev_zre = torch.where(torch.rand(density.shape, device="cuda", generator=g) < 0.1, 11.5, -1.0).float()
spec = RH.recomb_spec(n, model=2, cell_recomb=1, r_bubble_max=40.0)
kw = dict(n_ion=n_ion, whalo_sfr=whalo, prev_nrec=prev_nrec, prev_z_reion=prev_zre)
ntot = n**3


def timed(f, reps=3):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = f()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    return sorted(ts)[len(ts) // 2], r


out = {"hii_dim": n, "world": world, "n_radii": spec.n_radii}
t, (buf0, _, rep0) = timed(lambda: api.ionize_grids(spec, density, **kw))
out["single_pass_ms"] = round(t, 2)
fc = [torch.empty(ntot, dtype=torch.uint8, device="cuda") for _ in range(world)]
gg = [torch.empty(ntot, dtype=torch.float32, device="cuda") for _ in range(world)]
out["shard_phase_ms"] = []
for rank in range(world):
    t, _ = timed(lambda: api.ionize_shard_radii_rc(spec, rank, world, fc[rank], gg[rank], density, **kw))  # noqa: B023
    out["shard_phase_ms"].append(round(t, 2))
owner = D.owner_rank(spec.n_radii, world)
lo, hi = D.ts_slab(ntot, owner, world)
stride = (hi - lo + 15) // 16 * 16
pm = torch.zeros((world - 1, stride), dtype=torch.uint8, device="cuda")
pg = torch.zeros((world - 1, stride), dtype=torch.float32, device="cuda")
peers = [q for q in range(world) if q != owner]
for i, q in enumerate(peers):
    pm[i, : hi - lo], pg[i, : hi - lo] = fc[q][lo:hi], gg[q][lo:hi]
t, _ = timed(lambda: api.combine_cross_g12(fc[owner][lo:hi], gg[owner][lo:hi], pm, pg))
out["combine_slab_ms"] = round(t, 3)
m = torch.stack(fc).max(0).values
win = torch.stack(fc).argmax(0)
v = torch.stack(gg).gather(0, win[None])[0]
t, (buf, _, rep) = timed(lambda: api.ionize_shard_finish_rc(spec, m, v, density, **kw))
out["finish_ms"] = round(t, 2)
out["bit_identical"] = bool(torch.equal(buf.neutral_fraction, buf0.neutral_fraction)
                            and torch.equal(buf.ionisation_rate_G12, buf0.ionisation_rate_G12))
out["bytes_per_link_and_hop"] = 5 * ntot // world
keys = torch.zeros(ntot, dtype=torch.int64, device="cuda")
t, _ = timed(lambda: api.ionize_shard_radii_keys(spec, 0, world, keys, density, **kw))
out["key_shard_phase_rank0_ms"] = round(t, 2)
print(json.dumps(out))
