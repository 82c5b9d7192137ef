"""Multi-process (gloo, CPU, world 2 and 3) test of the exchange plan of the finish phase by cell slabs
(round 5; csrc/host/shard_rccl.c: slab_mask_exchange / slab_exchange_rccl, mirrored over
torch.distributed in 21cmfast_amd/distributed.py).

The HIP kernels cannot run here; what is under test is the part that is not GPU code: the slab deal
(c21cm_ionize_shard_slab, host arithmetic of the library), the all-to-all of the packed first
crossings by slab with the OR on arrival, and the all-gather of the chunk sums / flag / output slabs.
Reference: IonisationBox.c:1531-1588 ("largest radius that ionises the cell" is order independent),
:1031-1256 (the finish is per cell)."""

import importlib
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

D = importlib.import_module("21cmfast_amd.distributed")
W = importlib.import_module("21cmfast_amd.workloads")
api = importlib.import_module("21cmfast_amd.grid_api")


@pytest.mark.parametrize("n,nz", [(16, None), (64, None), (50, None), (128, 1024), (512, None), (1024, None)])
def test_slab_deal_covers_every_chunk_and_cell_once(n, nz):
    spec = W.ionize_spec(n, hii_dim_z=nz)
    ntot = n * n * (nz or n)
    for world in (1, 2, 3, 5, 8):
        slabs = [api.shard_slab(spec, r, world) for r in range(world)]
        assert slabs[0]["cell_begin"] == 0 and slabs[0]["chunk_begin"] == 0
        assert slabs[-1]["cell_end"] == ntot and slabs[-1]["chunk_end"] == slabs[0]["n_chunks"]
        for a, b in zip(slabs[:-1], slabs[1:]):
            assert a["cell_end"] == b["cell_begin"] and a["chunk_end"] == b["chunk_begin"]
            assert a["cell_end"] % 512 == 0  # whole words of the packed grid, 16-byte rows
        cc, nch = slabs[0]["chunk_cells"], slabs[0]["n_chunks"]
        assert (nch - 1) * cc < ntot <= nch * cc and nch <= 2048
        counts = [s["chunk_end"] - s["chunk_begin"] for s in slabs]
        assert max(counts) - min(counts) <= 1


def _worker(rank, world, port, n, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    spec = W.ionize_spec(n, r_bubble_max=8.0)
    ntot = n ** 3
    # this rank's first crossings: its own radii, on a pattern every process can recompute
    fc = _rank_mask(n, rank, world, spec.n_radii)
    t = torch.from_numpy(fc.copy())
    D.slab_mask_exchange(t, spec, rank, world)
    sl = api.shard_slab(spec, rank, world)
    # exchange 2 on synthetic chunk sums: chunk c carries (c + 1) / 7 from its owner
    nch = sl["n_chunks"]
    stars = torch.zeros(nch, dtype=torch.float64)
    xh = torch.zeros(nch, dtype=torch.float64)
    stars[sl["chunk_begin"]:sl["chunk_end"]] = torch.arange(sl["chunk_begin"], sl["chunk_end"], dtype=torch.float64) / 7 + 1
    xh[sl["chunk_begin"]:sl["chunk_end"]] = -torch.arange(sl["chunk_begin"], sl["chunk_end"], dtype=torch.float64) / 3 - 1
    flag = torch.tensor([1 if rank == world - 1 else 0], dtype=torch.int32)
    out = torch.full((ntot,), -5.0, dtype=torch.float32)
    out[sl["cell_begin"]:sl["cell_end"]] = float(rank + 1)
    D.slab_sums_exchange(stars, xh, flag, [out, None, None], spec, rank, world, gather_outputs=True)
    q.put((rank, t.numpy().reshape(-1)[sl["cell_begin"]:sl["cell_end"]].copy(), stars.numpy().copy(),
           xh.numpy().copy(), int(flag.item()), out.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


def _rank_mask(n, rank, world, n_radii):
    rng = np.random.default_rng(1000 + rank)
    fc = np.zeros(n ** 3, np.uint8)
    for r in D.radii_of_rank(n_radii, rank, world):
        hit = rng.random(n ** 3) < 0.03
        fc[(fc == 0) & hit] = r
    return fc


@pytest.mark.parametrize("world", [2, 3])
def test_slab_exchanges_over_gloo(world):
    n = 32
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        r, mask, stars, xh, flag, out = q.get(timeout=300)
        got[r] = (mask, stars, xh, flag, out)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    spec = W.ionize_spec(n, r_bubble_max=8.0)
    union = np.zeros(n ** 3, bool)
    for r in range(world):
        union |= _rank_mask(n, r, world, spec.n_radii) > 0
    assert 0.05 < union.mean() < 0.95
    slabs = [api.shard_slab(spec, r, world) for r in range(world)]
    nch = slabs[0]["n_chunks"]
    for r in range(world):
        mask, stars, xh, flag, out = got[r]
        sl = slabs[r]
        # exchange 1: the rank's slab holds the OR over all ranks (as 0 / 1 bytes)
        np.testing.assert_array_equal(mask, union[sl["cell_begin"]:sl["cell_end"]].astype(np.uint8))
        # exchange 2: every rank ends with every chunk's sums, the max flag and all output slabs
        np.testing.assert_array_equal(stars, np.arange(nch) / 7 + 1)
        np.testing.assert_array_equal(xh, -np.arange(nch) / 3 - 1)
        assert flag == 1
        for p in range(world):
            assert (out[slabs[p]["cell_begin"]:slabs[p]["cell_end"]] == p + 1).all()
