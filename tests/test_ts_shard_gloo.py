"""Multi-process (world_size = 2 and 3, gloo, CPU) test of the exchange plan of the sharded
ComputeTsBox: the shell deal, the cell slabs, the reduce-scatter of the per-cell partial sums and
the all-gather of the output boxes through torch.distributed (21cmfast_amd/distributed.py).  The
HIP phases cannot run here: each rank's partial sums are synthetic (a function of the rank's
shells), what is under test is everything that is not GPU code.  The C library (loaded on the CPU)
must agree on the slab boundaries."""

import ctypes as C
import importlib
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

D = importlib.import_module("21cmfast_amd.distributed")


def shell_term(shell, ntot, rows):
    """A deterministic stand-in for one shell's contribution to the sums of every cell."""
    i = np.arange(ntot, dtype=np.float64)
    return np.stack([np.sin(0.37 * (shell + 1) + 1e-3 * i * (k + 1)) * (1 + shell) for k in range(rows)])


def _worker(rank, world, port, ntot, n_step, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rows = 4
    partial = np.zeros((rows, ntot))
    for sh in D.shells_of_rank(n_step, rank, world):
        partial += shell_term(sh, ntot, rows)
    slab = D.ts_reduce_scatter(torch.from_numpy(partial), rank, world)
    b, e = D.ts_slab(ntot, rank, world)
    # the "temperature update" of the slab: any per-cell function of the complete sums
    box = torch.zeros(ntot, dtype=torch.float32)
    box[b:e] = (slab[0] + 2 * slab[1] - slab[2] * slab[3]).float()
    D.ts_all_gather(box, rank, world)
    q.put((rank, box.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_reduce_scatter_and_all_gather_reproduce_the_sequential_sums(world):
    ntot, n_step = 4 * 1031 + 2, 40  # a cell count that the slabs do not divide evenly
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, ntot, n_step, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    full = sum(shell_term(sh, ntot, 4) for sh in range(n_step))
    want = (full[0] + 2 * full[1] - full[2] * full[3]).astype(np.float32)
    for rank in range(world):
        np.testing.assert_allclose(got[rank], want, rtol=2e-6, atol=1e-5)  # every rank: the full box
    np.testing.assert_array_equal(got[0], got[world - 1])


def test_c_library_agrees_on_deal_and_slabs(pkg):
    lib = pkg.load()
    lib.c21cm_ts_slab_begin.restype = C.c_size_t
    lib.c21cm_ts_slab_begin.argtypes = [C.c_size_t, C.c_int, C.c_int]
    lib.c21cm_ts_shard_shells.restype = C.c_int
    for ntot in (512**3, 50**3, 4 * 1031 + 2):
        for world in (1, 2, 3, 8):
            for r in range(world):
                assert D.ts_slab(ntot, r, world) == (lib.c21cm_ts_slab_begin(ntot, world, r),
                                                     lib.c21cm_ts_slab_begin(ntot, world, r + 1))
    idx = (C.c_int * 64)()
    for n_step in (1, 7, 40):
        for world in (1, 2, 3, 8):
            for r in range(world):
                n = lib.c21cm_ts_shard_shells(n_step, r, world, idx)
                assert list(idx[:n]) == D.shells_of_rank(n_step, r, world)
