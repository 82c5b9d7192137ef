"""Multi-process (world_size = 2, gloo, CPU) test of the R-loop sharding plan.

The HIP kernels cannot run here, so each rank's shard phase is emulated with the CPU oracle
(one oracle pass per radius restricted to that radius); what is under test is the part that
is NOT GPU code: the radius -> rank deal, the owner rank, and the uint8 max-reduce through
torch.distributed producing exactly the sequential loop's first-crossing grid."""

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
S = importlib.import_module("21cmfast_amd.structs")


def test_radius_deal_covers_every_radius_once():
    for n_radii in (1, 2, 7, 30, 40):
        for world in (1, 2, 3, 4, 8):
            seen = []
            for rank in range(world):
                rr = D.radii_of_rank(n_radii, rank, world)
                assert rr == sorted(rr, reverse=True)
                seen += rr
            assert sorted(seen) == list(range(1, n_radii))
            owner = D.owner_rank(n_radii, world)
            counts = [len(D.radii_of_rank(n_radii, r, world)) for r in range(world)]
            assert counts[owner] == min(counts)
    assert D.radii_of_rank(40, 0, 8, r_lowest=10) == [39, 31, 23, 15]


def _single_radius_mask(oracle, spec, density, n_ion, r):
    """Cells whose barrier is crossed at radius index r alone (no partial ionisation)."""
    one = S.IonizeSpec.from_buffer_copy(spec)
    one.n_radii = 2
    one.r_lowest = 1
    one.R[0] = spec.R[0]
    one.R[1] = spec.R[r]
    one.sigma_maxmass[1] = spec.sigma_maxmass[r]
    out = oracle.ionize_grids(one, density, n_ion)
    return out["neutral_fraction"] == 0


def _worker(rank, world, port, n, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    oracle = importlib.import_module("oracle.oracle")
    oracle.set_threads(2)
    spec = W.ionize_spec(n, r_bubble_max=8.0)
    density = W.density_field_numpy(n, seed=17)
    n_ion = W.nion_from_density(density)
    first_cross = np.zeros((n, n, n), np.uint8)
    for r in D.radii_of_rank(spec.n_radii, rank, world):  # descending
        m = _single_radius_mask(oracle, spec, density, n_ion, r)
        first_cross[(first_cross == 0) & m] = r
    t = torch.from_numpy(first_cross)
    owner = D.owner_rank(spec.n_radii, world)
    D.reduce_first_cross(t, owner)
    if rank == owner:
        q.put((owner, t.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_reduce_reproduces_sequential_loop(oracle):
    n, world = 16, 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    owner, reduced = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    spec = W.ionize_spec(n, r_bubble_max=8.0)
    assert owner == D.owner_rank(spec.n_radii, world)
    density = W.density_field_numpy(n, seed=17)
    n_ion = W.nion_from_density(density)
    # sequential reference: all radii >= 1, largest first
    expect = np.zeros((n, n, n), np.uint8)
    for r in range(spec.n_radii - 1, 0, -1):
        m = _single_radius_mask(oracle, spec, density, n_ion, r)
        expect[(expect == 0) & m] = r
    np.testing.assert_array_equal(reduced, expect)
    # and the union equals the oracle's own loop over radii >= 1
    seq = S.IonizeSpec.from_buffer_copy(spec)
    seq.r_lowest = 1
    out = oracle.ionize_grids(seq, density, n_ion)
    np.testing.assert_array_equal(reduced > 0, out["neutral_fraction"] == 0)
    assert 0.02 < (reduced > 0).mean() < 0.98
