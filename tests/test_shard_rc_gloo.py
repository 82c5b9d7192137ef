"""Multi-process (world_size = 2 and 3, gloo, CPU) test of the exchange of the sharded FUSED
recombination loop (21cmfast_amd/distributed.py: cross_g12_exchange; the C library does the same
over RCCL point to point, shard_rccl.c: exchange_cross_g12): every rank holds a uint8
first-crossing index grid over ITS radii and Gamma_12 at those crossings; after the reduce-scatter
by cell slabs and the gather, the owner holds per cell the larger index with the Gamma_12 of the
rank that found it.  The HIP phases cannot run here: the per-rank grids are synthetic."""

import importlib
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

D = importlib.import_module("21cmfast_amd.distributed")


def rank_grids(rank, world, ntot, n_radii):
    """Synthetic shard-phase output: each cell crosses at one of the rank's radii or not at all;
    Gamma_12 is a function of (cell, index) so the test can tell whose value arrived."""
    rng = np.random.default_rng(1234 + rank)
    radii = D.radii_of_rank(n_radii, rank, world)
    radii = [r for r in radii if r > 0]
    pick = rng.integers(0, len(radii) + 2, ntot)
    mask = np.zeros(ntot, np.uint8)
    for i, r in enumerate(radii):
        mask[pick == i] = r
    g12 = np.where(mask > 0, 0.001 * np.arange(ntot) + mask.astype(np.float64), 0.0).astype(np.float32)
    return mask, g12


def _worker(rank, world, port, ntot, n_radii, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mask, g12 = rank_grids(rank, world, ntot, n_radii)
    owner = D.owner_rank(n_radii, world)
    fc, g = torch.from_numpy(mask), torch.from_numpy(g12)
    D.cross_g12_exchange(fc, g, rank, world, owner)
    if rank == owner:
        q.put((fc.numpy().copy(), g.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_owner_receives_the_larger_index_with_its_gamma12(world):
    ntot, n_radii = 4 * 1531 + 3, 23
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, ntot, n_radii, q)) for r in range(world)]
    for p in procs:
        p.start()
    fc, g = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    grids = [rank_grids(r, world, ntot, n_radii) for r in range(world)]
    masks = np.stack([m for m, _ in grids])
    vals = np.stack([v for _, v in grids])
    win = masks.argmax(0)
    np.testing.assert_array_equal(fc, masks.max(0))
    np.testing.assert_array_equal(g, vals[win, np.arange(ntot)])
    assert (masks.max(0) > 0).mean() > 0.5


def test_slab_bounds_match_the_c_library():
    lib = importlib.import_module("21cmfast_amd").load()
    import ctypes as C

    lib.c21cm_ts_slab_begin.restype = C.c_size_t
    lib.c21cm_ts_slab_begin.argtypes = [C.c_size_t, C.c_int, C.c_int]
    for ntot in (128**3, 4 * 1531 + 3, 7):
        for world in (1, 2, 3, 8):
            for r in range(world):
                assert D.ts_slab(ntot, r, world) == (lib.c21cm_ts_slab_begin(ntot, world, r),
                                                     lib.c21cm_ts_slab_begin(ntot, world, r + 1))
