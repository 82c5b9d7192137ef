"""Measured pieces of the multi-GPU budget of the sharded ComputeIonizedBox with the finish phase by cell
slabs (diagnostic, GPU box only): on ONE GPU, what rank `r` of a `world`-rank run does --
its shard phase (pre-loop + its radii), the pack of its first crossings, the OR-unpack of its slab, its
slab's final sweep + reduce -- each timed with HIP events, beside the single-GPU call and the owner-finish
pieces (whole-grid OR-unpack, whole final sweep).  The exchanges themselves need more than one GPU: their
bytes per link are printed.

    PYTHONPATH=. python tools/time_slab_finish.py [HII_DIM=1024] [world=8]
"""
import importlib
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

W = importlib.import_module("21cmfast_amd.workloads")
api = importlib.import_module("21cmfast_amd.grid_api")
D = importlib.import_module("21cmfast_amd.distributed")

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
spec = W.ionize_spec(n)
density = W.density_field_torch(n, seed=12345)
n_ion = W.nion_from_density(density)
ntot = n**3


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


buf = api.IonizeBuffers(density)
rep_single = {}


def single():
    buf.reset()
    _, _, rep_single["rep"] = api.ionize_grids(spec, density, n_ion, buffers=buf)


out = {"hii_dim": n, "world": world, "n_radii": spec.n_radii}
out["single_gpu_ms"] = timed(single, 2)
rs = rep_single["rep"]
out["single_gpu_phases_ms"] = {"preloop": rs.ms_preloop, "rloop": rs.ms_rloop, "postloop": rs.ms_postloop}
fc = torch.zeros((n, n, n), dtype=torch.uint8, device="cuda")
owner = D.owner_rank(spec.n_radii, world)
counts = [len(D.radii_of_rank(spec.n_radii, r, world)) for r in range(world)]
busiest = counts.index(max(counts))
out["radii_per_rank"] = counts
out["shard_phase_ms_busiest_rank"] = timed(
    lambda: api.ionize_shard_radii(spec, busiest, world, fc, density, n_ion, want_report=False), 2)
out["shard_phase_ms_owner_rank"] = timed(
    lambda: api.ionize_shard_radii(spec, owner, world, fc, density, n_ion, want_report=False), 2)
bits = {}
out["pack_mask_bits_ms"] = timed(lambda: bits.__setitem__("b", api.shard_pack_mask_bits(fc)), 5)
sl = api.shard_slab(spec, busiest, world)
w0, w1 = sl["cell_begin"] // 32, (sl["cell_end"] + 31) // 32
pieces = torch.stack([bits["b"][w0:w1] for _ in range(world)]).contiguous()
flat = fc.view(-1)
out["slab_or_unpack_ms"] = timed(
    lambda: api.shard_or_unpack_mask_bits(pieces, flat[sl["cell_begin"]:sl["cell_end"]]), 5)
whole = torch.stack([bits["b"] for _ in range(world)]).contiguous()
out["owner_or_unpack_whole_grid_ms"] = timed(lambda: api.shard_or_unpack_mask_bits(whole, flat), 3)
sbuf = api.IonizeBuffers(density)
out["slab_finish_ms"] = timed(
    lambda: api.ionize_shard_finish_slab(spec, flat, busiest, world, density, n_ion, buffers=sbuf), 3)
obuf = api.IonizeBuffers(density)
out["owner_finish_ms"] = timed(lambda: api.ionize_shard_finish(spec, fc, density, n_ion, buffers=obuf), 3)
words = (ntot + 31) // 32
out["exchange_bytes_per_link"] = {
    "slab_mask_bits": 4 * (w1 - w0), "owner_mask_bits": 4 * words,
    "chunk_sums": 16 * (sl["chunk_end"] - sl["chunk_begin"]),
    "output_slabs_all_gather_3_boxes": 12 * (sl["cell_end"] - sl["cell_begin"])}
crit = out["shard_phase_ms_busiest_rank"] + out["pack_mask_bits_ms"] + out["slab_or_unpack_ms"] + out["slab_finish_ms"]
out["critical_path_compute_ms"] = crit
out["speedup_bound_without_exchange"] = out["single_gpu_ms"] / crit
print(json.dumps(out))
