"""Diagnostic (GPU box): the 1024^3 single pass against its two-rank shard phases, repeated under contention;
on a mismatch print where the cells differ."""
import importlib, sys, os
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch
W = importlib.import_module("21cmfast_amd.workloads")
api = importlib.import_module("21cmfast_amd.grid_api")
n = int(os.environ.get("N", "1024"))
world = 2
spec = W.ionize_spec(n)
density = W.density_field_torch(n)
n_ion = W.nion_from_density(density)
buf, box, rep = api.ionize_grids(spec, density, n_ion)
x1 = buf.neutral_fraction.clone()
for it in range(int(os.environ.get("REPS", "6"))):
    masks = []
    for rank in range(world):
        fc = torch.zeros((n, n, n), dtype=torch.uint8, device="cuda")
        api.ionize_shard_radii(spec, rank, world, fc, density, n_ion, want_report=False)
        masks.append(fc)
    torch.cuda.synchronize()
    # each rank's mask against a second run of the same rank
    for rank in range(world):
        fc = torch.zeros((n, n, n), dtype=torch.uint8, device="cuda")
        api.ionize_shard_radii(spec, rank, world, fc, density, n_ion, want_report=False)
        torch.cuda.synchronize()
        d = (fc != masks[rank])
        nd = int(d.sum())
        if nd:
            idx = d.nonzero()
            print(f"iter {it} rank {rank}: shard phase NOT reproducible: {nd} cells; x {idx[:,0].min().item()}..{idx[:,0].max().item()} "
                  f"y {idx[:,1].min().item()}..{idx[:,1].max().item()} z {idx[:,2].min().item()}..{idx[:,2].max().item()}; "
                  f"values {sorted(set(fc[d].tolist()))[:8]} vs {sorted(set(masks[rank][d].tolist()))[:8]}", flush=True)
            xs = torch.bincount(idx[:, 0], minlength=n)
            print("   x planes with differences:", xs.nonzero().view(-1).tolist()[:40], flush=True)
            zs = torch.bincount(idx[:, 2], minlength=n)
            print("   z with differences:", zs.nonzero().view(-1).tolist()[:40], flush=True)
        del fc
    reduced = torch.maximum(masks[0], masks[1])
    buf.reset()
    buf, _, rep2 = api.ionize_shard_finish(spec, reduced, density, n_ion, buffers=buf)
    torch.cuda.synchronize()
    d = (x1 != buf.neutral_fraction)
    nd = int(d.sum())
    print(f"iter {it}: sharded vs single: {nd} cells differ", flush=True)
    if nd:
        idx = d.nonzero()
        print(f"   x {idx[:,0].min().item()}..{idx[:,0].max().item()} y {idx[:,1].min().item()}..{idx[:,1].max().item()} z {idx[:,2].min().item()}..{idx[:,2].max().item()}", flush=True)
        # which radius index do the differing cells carry in the reduced mask / single pass?
        print("   reduced mask values there:", torch.bincount(reduced[d].int(), minlength=41).nonzero().view(-1).tolist(), flush=True)
    # and the single pass again
    buf.reset()
    buf, _, _ = api.ionize_grids(spec, density, n_ion, buffers=buf)
    torch.cuda.synchronize()
    nd = int((x1 != buf.neutral_fraction).sum())
    if nd:
        print(f"iter {it}: SINGLE pass not reproducible: {nd} cells", flush=True)
    del masks, reduced
