"""R-loop sharding of ComputeIonizedBox over the GPUs of one node.

One process per GPU.  Every rank holds the (replicated) density / emissivity grids, filters
its share of the radii and records, per cell, the largest radius index whose barrier was
crossed (uint8, 0 = none).  ONE collective combines the shards: a max-reduce of that uint8
grid onto the rank that owns radius index 0, which then applies the mask, runs the
cell-scale radius (partial ionisation) and the post-loop.  "Largest radius that ionises the
cell" is order independent, so the result is identical to the sequential loop
(reference: src/py21cmfast/src/IonisationBox.c:1531-1588).

Two implementations of the exchange:
* ``sharded_ionize_c``  -- the product path: the reduce happens INSIDE the C library through its
  own RCCL communicator (csrc/host/shard_rccl.c, ``c21cm_ionize_sharded``); torch.distributed
  only carries the 128-byte unique id once (``grid_api.shard_init_from_torch``).  With the
  communicator initialised the drop-in ``ComputeIonizedBox`` shards by itself.
* ``sharded_ionize``    -- the same phases with the reduce through ``torch.distributed``
  (backend "gloo" in the CPU tests and when several ranks share one GPU, where RCCL refuses).
"""

from __future__ import annotations


def radii_of_rank(n_radii: int, rank: int, world: int, r_lowest: int = 0) -> list[int]:
    """Radius indices (descending, all >= 1) processed by `rank` in the shard phase:
    indices n_radii-1 ... 1 dealt round-robin, largest first (matches
    c21cm_ionize_shard_radii in csrc/host/ionize_driver.c)."""
    out = []
    r = n_radii - 1 - rank
    while r >= 1 and r >= r_lowest:
        out.append(r)
        r -= world
    return out


def owner_rank(n_radii: int, world: int) -> int:
    """The rank that the round-robin deal would hand radius index 0 to: it has the fewest
    shard radii, so it also runs the finish step and is the destination of the reduce."""
    return (n_radii - 1) % world


def reduce_first_cross(first_cross, owner: int, group=None):
    """Max-reduce the uint8 first-crossing grids onto `owner` (in place on that rank)."""
    import torch.distributed as dist

    dist.reduce(first_cross, dst=owner, op=dist.ReduceOp.MAX, group=group)
    return first_cross


def sharded_ionize(spec, density, n_ion, buffers, first_cross, rank: int, world: int, group=None,
                   collect_means: bool | None = None):
    """One sharded ComputeIonizedBox pass; returns the report on the owner rank, else None.

    ``collect_means``: also combine the per-radius f_coll grid means of all ranks (a 2 KB
    sum-reduce; costs one host synchronisation per rank) and hand them to the finish step.
    Default: only when the result depends on them -- a Lagrangian model (``fix_mean == 0``)
    whose loop stops above index 0 reports the mean of radius ``r_lowest`` as
    ``box.mean_f_coll`` (reference: IonisationBox.c:1623-1628)."""
    from . import grid_api as api

    owner = owner_rank(spec.n_radii, world)
    if collect_means is None:
        collect_means = (not spec.fix_mean) and spec.r_lowest > 0
    rep = api.ionize_shard_radii(spec, rank, world, first_cross, density, n_ion,
                                 want_report=bool(collect_means))
    if collect_means:
        import torch
        import torch.distributed as dist

        means = torch.tensor(list(rep.f_coll_grid_mean)[: spec.n_radii], dtype=torch.float64,
                             device=first_cross.device)
        if world > 1:
            dist.reduce(means, dst=owner, op=dist.ReduceOp.SUM, group=group)
        if rank == owner:
            api.ionize_shard_set_means(means.cpu().numpy())
    reduce_first_cross(first_cross, owner, group)
    if rank == owner:
        _, _, rep = api.ionize_shard_finish(spec, first_cross, density, n_ion, buffers=buffers)
        return rep
    return None


def sharded_ionize_c(spec, density, n_ion, buffers, rank: int, world: int, **kw):
    """One sharded pass through c21cm_ionize_sharded; returns the report on the owner, else None."""
    from . import grid_api as api

    _, _, rep = api.ionize_sharded(spec, density, n_ion, buffers=buffers, **kw)
    return rep if rank == owner_rank(spec.n_radii, world) else None
