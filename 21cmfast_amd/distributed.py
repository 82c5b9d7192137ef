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


def slab_mask_exchange(first_cross, spec, rank: int, world: int, group=None):
    """Exchange 1 of the slab finish through torch.distributed (what c21cm_ionize_sharded does over RCCL
    point to point): every rank packs its uint8 first crossings to one bit per cell, sends peer p the
    words of p's slab and ORs what it receives into the bytes of its own slab (in place)."""
    import torch
    from . import grid_api as api

    fc = first_cross.view(-1)
    on_gpu = fc.is_cuda
    slabs = [api.shard_slab(spec, p, world) for p in range(world)]
    if on_gpu:
        bits = api.shard_pack_mask_bits(fc)
    else:  # CPU plan test (gloo): the same packing in torch
        pad = (-fc.numel()) % 32
        b = torch.nn.functional.pad((fc != 0).to(torch.int64), (0, pad)).view(-1, 32)
        w = (b << torch.arange(32, dtype=torch.int64)).sum(dim=1)
        bits = torch.where(w >= 2 ** 31, w - 2 ** 32, w).to(torch.int32)

    def words(p):
        return slabs[p]["cell_begin"] // 32, (slabs[p]["cell_end"] + 31) // 32

    import torch.distributed as dist

    lo, hi = words(rank)
    # (gloo moves CPU tensors: ranks sharing one GPU in the plumbing test stage through the host)
    xdev = "cpu" if dist.get_backend(group) == "gloo" else fc.device
    recv = [torch.empty(hi - lo, dtype=torch.int32, device=xdev) for _ in range(world)]
    send = [bits[slice(*words(p))].contiguous().to(xdev) for p in range(world)]
    _all_to_all_p2p(recv, send, rank, world, group)
    mine = fc[slabs[rank]["cell_begin"]:slabs[rank]["cell_end"]]
    pieces = torch.stack(recv).contiguous().to(fc.device)
    if on_gpu:
        api.shard_or_unpack_mask_bits(pieces, mine)
    else:
        v = pieces[0].clone()
        for q in range(1, world):
            v |= pieces[q]
        cells = ((v.to(torch.int64).view(-1, 1) >> torch.arange(32, dtype=torch.int64)) & 1).view(-1)
        mine.copy_(cells[: mine.numel()].to(torch.uint8))
    return first_cross


def slab_sums_exchange(stars, xh, flag, outs, spec, rank: int, world: int, group=None,
                       gather_outputs: bool = False):
    """Exchange 2 of the slab finish: all-gather of the chunk partial sums (`stars`, `xh`: float64
    [n_chunks], this rank's chunk range filled), max of the non-finite flag, and -- only with
    ``gather_outputs`` -- the all-gather of the output slabs (`outs`: flat float32 grids), all in
    place."""
    import torch
    import torch.distributed as dist
    from . import grid_api as api

    slabs = [api.shard_slab(spec, p, world) for p in range(world)]
    for t in (stars, xh):
        mine = t[slabs[rank]["chunk_begin"]:slabs[rank]["chunk_end"]].clone()
        reqs, bufs = [], {}
        for p in range(world):
            if p == rank:
                continue
            bufs[p] = torch.empty(slabs[p]["chunk_end"] - slabs[p]["chunk_begin"], dtype=t.dtype,
                                  device=t.device)
            reqs.append(dist.isend(mine, dst=p, group=group))
            reqs.append(dist.irecv(bufs[p], src=p, group=group))
        for q in reqs:
            q.wait()
        for p, b in bufs.items():
            t[slabs[p]["chunk_begin"]:slabs[p]["chunk_end"]] = b
    if world > 1:
        dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group)
    if gather_outputs:
        for o in outs:
            if o is None:
                continue
            mine = o[slabs[rank]["cell_begin"]:slabs[rank]["cell_end"]].clone()
            reqs, bufs = [], {}
            for p in range(world):
                if p == rank:
                    continue
                bufs[p] = torch.empty(slabs[p]["cell_end"] - slabs[p]["cell_begin"], dtype=o.dtype,
                                      device=o.device)
                reqs.append(dist.isend(mine, dst=p, group=group))
                reqs.append(dist.irecv(bufs[p], src=p, group=group))
            for q in reqs:
                q.wait()
            for p, b in bufs.items():
                o[slabs[p]["cell_begin"]:slabs[p]["cell_end"]] = b


def sharded_ionize_slabs(spec, density, n_ion, buffers, first_cross, rank: int, world: int, group=None,
                         gather_outputs: bool = False, **kw):
    """One sharded ComputeIonizedBox pass with the finish phase by cell slabs and both exchanges through
    torch.distributed (the C phases are the ones c21cm_ionize_sharded runs).  Every rank returns the
    report -- the scalars are complete everywhere; the outputs are the rank's slab unless
    ``gather_outputs``."""
    from . import grid_api as api

    api.ionize_shard_radii(spec, rank, world, first_cross, density, n_ion, want_report=False, **kw)
    slab_mask_exchange(first_cross, spec, rank, world, group)

    def exchange(st, local_status):
        import torch
        import torch.distributed as dist

        gloo = dist.get_backend(group) == "gloo"
        bad = torch.tensor([1 if local_status else 0], dtype=torch.int32,
                           device="cpu" if gloo else first_cross.device)
        if world > 1:
            dist.all_reduce(bad, op=dist.ReduceOp.MAX, group=group)
        if int(bad.item()):
            return local_status or 3
        stars = api.device_view(st.partials_stars, st.n_chunks, "f8")
        xh = api.device_view(st.partials_xh, st.n_chunks, "f8")
        flag = api.device_view(st.flag, 1, "i4")
        outs = [api.device_view(st.out[i], st.ntot, "f4") if st.out[i] else None for i in range(3)]
        if gloo:  # host-staged: gloo moves CPU tensors
            hs, hx, hf = stars.cpu(), xh.cpu(), flag.cpu()
            ho = [o.cpu() if (o is not None and gather_outputs) else None for o in outs]
            slab_sums_exchange(hs, hx, hf, ho, spec, rank, world, group, gather_outputs)
            stars.copy_(hs), xh.copy_(hx), flag.copy_(hf)
            for o, h in zip(outs, ho):
                if h is not None:
                    o.copy_(h)
        else:
            slab_sums_exchange(stars, xh, flag, outs, spec, rank, world, group, gather_outputs)
        return 0

    _, _, rep = api.ionize_shard_finish_slab(spec, first_cross, rank, world, density, n_ion,
                                             buffers=buffers, exchange=exchange,
                                             outputs_gathered=gather_outputs, **kw)
    return rep


def sharded_ionize_c(spec, density, n_ion, buffers, rank: int, world: int, **kw):
    """One sharded pass through c21cm_ionize_sharded; returns the report on the owner, else None."""
    from . import grid_api as api

    _, _, rep = api.ionize_sharded(spec, density, n_ion, buffers=buffers, **kw)
    return rep if rank == owner_rank(spec.n_radii, world) else None


# ---- ComputeTsBox: the N_STEP_TS shells dealt over the ranks (csrc/host/abi_compute.c: ts_box_run) ----
def shells_of_rank(n_step: int, rank: int, world: int) -> list[int]:
    """Shell indices (descending) of `rank`: n_step-1-rank, -world, ... (c21cm_ts_shard_shells)."""
    return list(range(n_step - 1 - rank, -1, -world))


def ts_slab(ntot: int, rank: int, world: int) -> tuple[int, int]:
    """Cell range [begin, end) whose temperature update `rank` runs (c21cm_ts_slab_begin: slab
    boundaries at multiples of 4 cells, the last slab ends at ntot)."""
    def begin(r):
        return ntot if r >= world else (ntot // 4 * r // world) * 4
    return begin(rank), begin(rank + 1)


def ts_reduce_scatter(partial, rank: int, world: int, group=None):
    """The exchange of the sharded ComputeTsBox through torch.distributed: `partial` [rows][N]
    (this rank's shells' sums); returns the complete sums of this rank's slab [rows][len], the
    ranks added in rank order (what c21cm_ts_box_sharded does over RCCL point to point)."""
    import torch
    import torch.distributed as dist

    rows, ntot = partial.shape
    mine = ts_slab(ntot, rank, world)
    recv = [torch.empty((rows, mine[1] - mine[0]), dtype=partial.dtype, device=partial.device)
            for _ in range(world)]
    send = [partial[:, slice(*ts_slab(ntot, p, world))].contiguous() for p in range(world)]
    dist.all_to_all(recv, send, group=group) if dist.get_backend(group) != "gloo" else _all_to_all_p2p(
        recv, send, rank, world, group)
    out = torch.zeros_like(recv[0])
    for r in range(world):  # rank order: the result does not depend on who computes it
        out += recv[r]
    return out


def _all_to_all_p2p(recv, send, rank, world, group):
    """gloo has no all_to_all on every build: the same exchange with isend / irecv."""
    import torch.distributed as dist

    recv[rank].copy_(send[rank])
    reqs = []
    for p in range(world):
        if p == rank:
            continue
        reqs.append(dist.isend(send[p], dst=p, group=group))
        reqs.append(dist.irecv(recv[p], src=p, group=group))
    for q in reqs:
        q.wait()


def ts_all_gather(box, rank: int, world: int, group=None):
    """Every rank's slab of an output box (flat view of length N, filled on [slab) by its owner)
    to everybody, in place."""
    import torch.distributed as dist

    ntot = box.numel()
    flat = box.view(-1)
    reqs = []
    b, e = ts_slab(ntot, rank, world)
    mine = flat[b:e].clone()
    bufs = {}
    for p in range(world):
        if p == rank:
            continue
        pb, pe = ts_slab(ntot, p, world)
        bufs[p] = flat[pb:pe].clone()
        reqs.append(dist.isend(mine, dst=p, group=group))
        reqs.append(dist.irecv(bufs[p], src=p, group=group))
    for q in reqs:
        q.wait()
    for p, t in bufs.items():
        pb, pe = ts_slab(ntot, p, world)
        flat[pb:pe] = t
    return box


def cross_g12_exchange(first_cross, g12, rank: int, world: int, owner: int, group=None):
    """The exchange of the sharded FUSED recombination loop through torch.distributed (what
    c21cm_ionize_sharded does over RCCL point to point): per cell the entry of the rank with the
    larger first-crossing index wins, with ITS Gamma_12.  Hop 1: reduce-scatter by cell slabs
    (``ts_slab`` bounds) and combine; hop 2: the owner receives every combined slab in place.
    `first_cross` uint8 [N], `g12` float32 [N] (flat views; modified in place on the owner)."""
    import torch
    import torch.distributed as dist

    ntot = first_cross.numel()
    fc, g = first_cross.view(-1), g12.view(-1)
    lo, hi = ts_slab(ntot, rank, world)
    for t in (fc, g):
        recv = [torch.empty(hi - lo, dtype=t.dtype, device=t.device) for _ in range(world)]
        send = [t[slice(*ts_slab(ntot, p, world))].contiguous() for p in range(world)]
        _all_to_all_p2p(recv, send, rank, world, group)
        if t is fc:
            masks = recv
        else:
            vals = recv
    m, v = masks[rank].clone(), vals[rank].clone()
    for q in range(world):
        win = masks[q] > m
        m = torch.where(win, masks[q], m)
        v = torch.where(win, vals[q], v)
    fc[lo:hi], g[lo:hi] = m, v
    reqs = []
    if rank == owner:
        bufs = {}
        for p in range(world):
            if p == owner:
                continue
            a, b = ts_slab(ntot, p, world)
            if b > a:
                bufs[p] = (torch.empty(b - a, dtype=fc.dtype, device=fc.device),
                           torch.empty(b - a, dtype=g.dtype, device=g.device))
                reqs += [dist.irecv(bufs[p][0], src=p, group=group), dist.irecv(bufs[p][1], src=p, group=group)]
        for q in reqs:
            q.wait()
        for p, (bm, bg) in bufs.items():
            a, b = ts_slab(ntot, p, world)
            fc[a:b], g[a:b] = bm, bg
    elif hi > lo:
        reqs = [dist.isend(m, dst=owner, group=group), dist.isend(v, dst=owner, group=group)]
        for q in reqs:
            q.wait()
    return first_cross, g12
