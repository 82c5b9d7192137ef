"""GPU parity: ComputeIonizedBox with a recombination model on the MI355X vs the CPU oracle
(reference: IonisationBox.c:583-663 N_rec / whalo_sfr grids, :1084-1140 barrier, Gamma_12, mean
free path, :1258-1340 set_recombination_rates; recombinations.c:64-92 the splined rate).

Tolerances as in test_gpu_ionize.py for x_HI / z_reion / T_k; on cells whose flag agrees, Gamma_12
rtol 1e-4 (it is R * prefactor * a filtered grid value), the mean free path exact (it is one of the
radii), N_rec rtol 2e-4 + atol 1e-7 (a spline of ln Gamma_12 times 1 - x_HI).
"""

import importlib

import numpy as np
import pytest

from recomb_helpers import inputs, recomb_spec
from test_gpu_ionize import compare

pytestmark = pytest.mark.gpu
W = importlib.import_module("21cmfast_amd.workloads")


@pytest.fixture(scope="module")
def api(gpu_lib):
    return importlib.import_module("21cmfast_amd.grid_api")


def run(api, spec, d, device, lagrangian, ts):
    kw = dict(prev_nrec=d["prev_nrec"], prev_z_reion=d["prev_z_reion"])
    if lagrangian:
        kw.update(n_ion=d["n_ion"], whalo_sfr=d["whalo_sfr"])
    if ts:
        kw.update(xe=d["xe"], Tneutral=d["Tneutral"])
    dens = d["density"]
    if device:
        import torch

        dens = torch.from_numpy(dens).cuda()
        kw = {k: torch.from_numpy(v).cuda() for k, v in kw.items()}
    buf, box, rep = api.ionize_grids(spec, dens, **kw)
    names = ("neutral_fraction", "z_reion", "kinetic_temperature", "ionisation_rate_G12",
             "mean_free_path", "cumulative_recombinations", "unnormalised_nion")
    out = {}
    for k in names:
        a = getattr(buf, k)
        if a is not None:
            out[k] = a.cpu().numpy() if device else a
    out["report"] = rep
    return out


CASES = [
    # n or shape, model, cell_recomb, lagrangian, ts, device
    (32, 2, 1, True, False, False),
    (64, 2, 0, True, False, True),     # native passes, N_rec and whalo_sfr grids filtered
    (40, 2, 0, False, False, False),   # rocFFT path, Eulerian closed form, N_rec filtered
    (64, 2, 1, False, False, True),
    ((32, 32, 64), 2, 0, True, True, True),   # + x_e grid of a spin-temperature run
    (32, 1, 1, True, False, True),     # homogeneous: one number
    (35, 1, 1, False, False, False),
    # 256-point z-lines, windows evaluated in pass X: the FUSED recombination loop (round 3)
    ((128, 128, 256), 2, 1, True, False, True),
    ((128, 128, 256), 1, 1, True, False, True),
]


@pytest.mark.parametrize("n,model,cell,lagrangian,ts,device", CASES)
def test_recombination_models_match_oracle(api, oracle, n, model, cell, lagrangian, ts, device):
    shape = (n, n, n) if isinstance(n, int) else n
    spec = recomb_spec(shape[0], model=model, cell_recomb=cell, lagrangian=lagrangian,
                       hii_dim_z=shape[2], ts=int(ts))
    d = inputs(shape, seed=shape[0] + 3 * model + cell, ts=ts)
    if model == 1:
        d["prev_nrec"] = np.full((1, 1, 1), 0.25, np.float32)
    okw = dict(prev_nrec=d["prev_nrec"], prev_z_reion=d["prev_z_reion"])
    if lagrangian:
        okw.update(n_ion=d["n_ion"], whalo_sfr=d["whalo_sfr"])
    else:
        okw.update(need_nion=True)
    if ts:
        okw.update(xe=d["xe"], Tneutral=d["Tneutral"])
    ref = oracle.ionize_grids(spec, d["density"], **okw)
    got = run(api, spec, d, device, lagrangian, ts)
    # "crossed a barrier in this call" = the mean free path was recorded (x_HI = 0 alone can also
    # come from the clamp of a partial ionisation once recombinations raise the barrier)
    ion_g, ion_r = got["mean_free_path"] > 0, ref["mean_free_path"] > 0
    compare(got, ref, spec, flags=(ion_g, ion_r))
    assert 0.03 < ion_r.mean() < 0.97
    same = ion_g == ion_r
    big = ion_r.size > 10**6
    if big:
        # 4 M cells: the float transforms leave ~3e-7 of a field's rms as noise (device and oracle
        # alike, tools/scratch/diag_win2.py), which is 1e-5 .. 1e-4 of the LOCAL emissivity in the
        # voids; there a cell within that of the barrier crosses one radius earlier or later, and
        # Gamma_12 -- a filtered grid value over (1 + delta) -- inherits the local relative noise.
        # Same bound as the ionisation flags for the crossing radius, then Gamma_12 where it agrees.
        same_R = same & (got["mean_free_path"] == ref["mean_free_path"])
        assert np.mean(~same_R & same) <= 2e-4
        same = same_R
    else:  # the small boxes: at most a couple of cells on a barrier cross one radius apart
        same_R = same & (got["mean_free_path"] == ref["mean_free_path"])
        assert np.sum(~same_R & same) <= 2
        same = same_R
    np.testing.assert_allclose(got["ionisation_rate_G12"][same], ref["ionisation_rate_G12"][same],
                               rtol=2e-3 if big else 1e-4,
                               atol=2e-6 * float(ref["ionisation_rate_G12"].max()) if big else 1e-9)
    np.testing.assert_array_equal(got["mean_free_path"][same], ref["mean_free_path"][same])
    assert (ref["ionisation_rate_G12"] > 0).any()
    if model == 2:
        np.testing.assert_allclose(got["cumulative_recombinations"][same],
                                   ref["cumulative_recombinations"][same], rtol=2e-4, atol=1e-7)
        assert (ref["cumulative_recombinations"] > d["prev_nrec"]).any()
    else:
        assert got["cumulative_recombinations"].shape == (1, 1, 1)
        assert float(got["cumulative_recombinations"].ravel()[0]) == pytest.approx(
            float(ref["cumulative_recombinations"].ravel()[0]), rel=1e-5)


def test_recombination_requests_are_validated(api):
    spec = recomb_spec(16, model=2)
    d = inputs((16, 16, 16))
    with pytest.raises(Exception, match="whalo_sfr"):
        api.ionize_grids(spec, d["density"], d["n_ion"], prev_nrec=d["prev_nrec"],
                         prev_z_reion=d["prev_z_reion"])
    with pytest.raises(Exception, match="cumulative_recombinations"):
        api.ionize_grids(spec, d["density"], d["n_ion"], whalo_sfr=d["whalo_sfr"],
                         prev_z_reion=d["prev_z_reion"])
    bad = recomb_spec(16, model=1, cell_recomb=0)
    with pytest.raises(Exception, match="CELL_RECOMB"):
        api.ionize_grids(bad, d["density"], d["n_ion"], whalo_sfr=d["whalo_sfr"],
                         prev_nrec=np.zeros((1, 1, 1), np.float32),
                         prev_z_reion=d["prev_z_reion"])


@pytest.mark.parametrize("lagrangian,cell", [(True, 0), (False, 1)])
def test_sharded_key_phases_equal_single_pass(api, lagrangian, cell):
    """R-loop sharding with a recombination model: each rank's first crossings travel as 64-bit
    (mean free path, Gamma_12) keys, the max over ranks (emulated here in one process for world =
    2 and 3) picks the largest ionising radius with ITS Gamma_12, and the finish phase must
    reproduce the single pass bit for bit."""
    import torch

    n = 64
    spec = recomb_spec(n, model=2, cell_recomb=cell, lagrangian=lagrangian)
    d = {k: torch.from_numpy(v).cuda() for k, v in inputs((n, n, n), seed=21).items()}
    kw = dict(prev_nrec=d["prev_nrec"], prev_z_reion=d["prev_z_reion"])
    if lagrangian:
        kw.update(n_ion=d["n_ion"], whalo_sfr=d["whalo_sfr"])
    buf0, _, rep0 = api.ionize_grids(spec, d["density"], **kw)
    torch.cuda.synchronize()
    assert 0.03 < float((buf0.mean_free_path > 0).float().mean()) < 0.97
    for world in (1, 2, 3):
        reduced = None
        for rank in range(world):
            keys = torch.zeros((n, n, n), dtype=torch.int64, device="cuda")
            api.ionize_shard_radii_keys(spec, rank, world, keys, d["density"], **kw)
            reduced = keys if reduced is None else torch.maximum(reduced, keys)
        buf, _, rep = api.ionize_shard_finish_keys(spec, reduced, d["density"], **kw)
        torch.cuda.synchronize()
        for name in ("neutral_fraction", "z_reion", "kinetic_temperature", "ionisation_rate_G12",
                     "mean_free_path", "cumulative_recombinations"):
            assert torch.equal(getattr(buf0, name), getattr(buf, name)), (world, name)
        assert rep.global_xH == rep0.global_xH


@pytest.mark.parametrize("recomb", [False, True])
def test_c_level_sharding_on_a_one_rank_communicator(api, recomb):
    """c21cm_ionize_sharded: shard phase, ncclReduce (RCCL resolved with dlopen, here on a one-rank
    communicator created through c21cm_shard_unique_id / c21cm_shard_init), finish phase and the
    output broadcast -- everything the 8-GPU run executes except a second participant."""
    import torch

    n = 64
    if recomb:
        spec = recomb_spec(n, model=2, cell_recomb=0)
        d = {k: torch.from_numpy(v).cuda() for k, v in inputs((n, n, n), seed=8).items()}
        kw = dict(n_ion=d["n_ion"], whalo_sfr=d["whalo_sfr"], prev_nrec=d["prev_nrec"],
                  prev_z_reion=d["prev_z_reion"])
        density = d["density"]
    else:
        spec = W.ionize_spec(n, r_bubble_max=12.0)
        density = torch.from_numpy(W.density_field_numpy(n, seed=3)).cuda()
        kw = dict(n_ion=W.nion_from_density(density))
    buf0, _, rep0 = api.ionize_grids(spec, density, **kw)
    api.shard_init_single()
    try:
        buf1, box1, rep1 = api.ionize_sharded(spec, density, broadcast=True, **kw)
        torch.cuda.synchronize()
    finally:
        api.shard_finalize()
    names = ["neutral_fraction", "z_reion", "kinetic_temperature"]
    if recomb:
        names += ["ionisation_rate_G12", "mean_free_path", "cumulative_recombinations"]
    for name in names:
        assert torch.equal(getattr(buf0, name), getattr(buf1, name)), name
    assert rep1.global_xH == rep0.global_xH
    assert 0.03 < float((buf0.neutral_fraction == 0).float().mean()) < 0.97


@pytest.mark.parametrize("recomb", [False, True])
@pytest.mark.parametrize("world", [2, 3, 8])
def test_c_level_sharding_with_an_emulated_transport(api, recomb, world):
    """c21cm_ionize_sharded for world = 2, 3, 8 with the transport replaced by an in-process
    mailbox (c21cm_shard_emulate): the ranks run one after the other, non-owners first.  Covers
    what a one-GPU box cannot run through RCCL: the radius deal, the per-rank slots of the 1-bit
    gather, the 64-bit key reduce, the owner's finish phase -- bit-identical to the single pass."""
    import torch

    D = importlib.import_module("21cmfast_amd.distributed")
    n = 64
    if recomb:
        spec = recomb_spec(n, model=2, cell_recomb=1)
        d = {k: torch.from_numpy(v).cuda() for k, v in inputs((n, n, n), seed=5).items()}
        kw = dict(n_ion=d["n_ion"], whalo_sfr=d["whalo_sfr"], prev_nrec=d["prev_nrec"],
                  prev_z_reion=d["prev_z_reion"])
        density = d["density"]
    else:
        spec = W.ionize_spec(n, r_bubble_max=20.0)
        density = torch.from_numpy(W.density_field_numpy(n, seed=13)).cuda()
        kw = dict(n_ion=W.nion_from_density(density))
    buf0, _, rep0 = api.ionize_grids(spec, density, **kw)
    owner = D.owner_rank(spec.n_radii, world)
    mailbox = torch.zeros(world * (n**3 // 8 + 64) + 8 * n**3, dtype=torch.uint8, device="cuda")
    try:
        for exchange in ("gather", "reduce"):
            if recomb and exchange == "reduce":
                continue  # recombination models always reduce their keys
            import os

            os.environ["C21CM_SHARD_EXCHANGE"] = exchange
            mailbox.zero_()
            result = None
            for rank in [r for r in range(world) if r != owner] + [owner]:
                api.shard_emulate(rank, world, mailbox)
                buf, _, rep = api.ionize_sharded(spec, density, **kw)
                if rank == owner:
                    result = (buf, rep)
            torch.cuda.synchronize()
            buf, rep = result
            names = ["neutral_fraction", "z_reion", "kinetic_temperature"]
            if recomb:
                names += ["ionisation_rate_G12", "mean_free_path", "cumulative_recombinations"]
            for name in names:
                assert torch.equal(getattr(buf0, name), getattr(buf, name)), (world, exchange, name)
            assert rep.global_xH == rep0.global_xH
    finally:
        import os

        os.environ.pop("C21CM_SHARD_EXCHANGE", None)
        api.shard_finalize()
    assert 0.03 < float((buf0.neutral_fraction == 0).float().mean()) < 0.97


@pytest.mark.parametrize("model,ts,cell_recomb,n", [(2, False, 1, 256), (1, False, 1, 256), (2, True, 1, 256),
                                                    (1, True, 1, 256), (2, False, 0, 256), (2, True, 0, 512),
                                                    (2, True, 1, 512)])
def test_fused_recombination_loop_equals_the_unfused_sequence(api, monkeypatch, model, ts, cell_recomb, n):
    """CELL_RECOMB runs ride the fused loop (whalo_sfr as a third spectrum of the wave-level pass Z,
    (1 + N_rec / (1 + delta)) in the barrier, Gamma_12 at first crossings, the mean free path from
    the first-crossing index); C21CM_RECOMB_FUSED=0 is the per-radius sequence of round 2.  Same
    crossings (up to cells within float round-off of the barrier), same Gamma_12 / N_rec."""
    import torch

    # ts (round 4): with the x_e grid of a spin-temperature run the filtered x_e is a third line of the
    # barrier kernel, f zeta > (1 - x_e)(1 + rec) (IonisationBox.c:1084-1118)
    # cell_recomb = 0 (round 4): N_rec of the previous snapshot filtered at the radius takes the third
    # line, f zeta > 1 + max(N_rec(R), 0) / (1 + delta_R) (IonisationBox.c:583-663,808-809,1093)
    # cell_recomb = 0 WITH an x_e grid (round 5, the last variant): four spectra, the N_rec transform parked
    # in LDS where the dense N_rec rows of CELL_RECOMB wait (512-point z-lines)
    spec = recomb_spec(n, model=model, cell_recomb=cell_recomb, r_bubble_max=20.0, ts=int(ts))
    d = {k: torch.from_numpy(v).cuda() for k, v in inputs((n, n, n), seed=77, ts=ts).items()}
    if model == 1:
        d["prev_nrec"] = torch.full((1, 1, 1), 0.25, dtype=torch.float32, device="cuda")
    kw = dict(n_ion=d["n_ion"], whalo_sfr=d["whalo_sfr"], prev_nrec=d["prev_nrec"],
              prev_z_reion=d["prev_z_reion"])
    if ts:
        kw.update(xe=d["xe"], Tneutral=d["Tneutral"])
    monkeypatch.setenv("C21CM_RECOMB_FUSED", "0")
    b0, _, r0 = api.ionize_grids(spec, d["density"], **kw)
    assert api.ionize_last_loop_flags() & 3 == 0
    monkeypatch.delenv("C21CM_RECOMB_FUSED")
    b1, _, r1 = api.ionize_grids(spec, d["density"], **kw)
    flags = api.ionize_last_loop_flags()
    assert flags & 2, flags  # the fused recombination loop ran
    assert bool(flags & 4) == (ts or not cell_recomb) and bool(flags & 16) == (ts and not cell_recomb), flags
    torch.cuda.synchronize()
    c0, c1 = b0.mean_free_path > 0, b1.mean_free_path > 0
    mism = float((c0 != c1).float().mean())
    assert mism <= 1e-5, mism
    # the crossing RADIUS of a cell within float noise of a barrier may differ by one step (two
    # transform pipelines, ~3e-7 of a field's rms apart: 1e-5 .. 1e-4 of the local value in voids)
    same = (c0 == c1) & (b0.mean_free_path == b1.mean_free_path)
    assert float((~same).float().mean()) <= 2e-4
    assert torch.equal(b0.z_reion[same], b1.z_reion[same])
    a, b = b0.neutral_fraction[same], b1.neutral_fraction[same]
    assert float((a - b).abs().max()) <= 6e-6
    a, b = b0.ionisation_rate_G12[same], b1.ionisation_rate_G12[same]
    assert bool(((a - b).abs() <= 1e-3 * a.abs() + 1e-5 * float(a.max())).all())
    if model == 2:
        a, b = b0.cumulative_recombinations[same], b1.cumulative_recombinations[same]
        assert float((a - b).abs().max()) <= 1e-4 * float(a.abs().max())
    assert abs(r0.global_xH - r1.global_xH) < 1e-6
    n_r = spec.n_radii
    np.testing.assert_allclose(np.array(r1.f_coll_grid_mean[:n_r]), np.array(r0.f_coll_grid_mean[:n_r]),
                               rtol=1e-6)
    assert 0.03 < float(c1.float().mean()) < 0.97


@pytest.mark.parametrize("model,ts,cell_recomb,n", [(2, False, 1, 256), (1, False, 1, 256), (2, True, 1, 256),
                                                    (2, False, 0, 256), (2, True, 0, 512)])
def test_sharded_fused_recombination_phases_equal_single_pass(api, model, ts, cell_recomb, n):
    """The fused recombination loop sharded (round 3): a rank's radii leave the uint8 first-crossing
    index + Gamma_12 (5 bytes per cell instead of the 8-byte keys); per cell the rank with the larger
    index wins with ITS Gamma_12 (c21cm_shard_combine_cross_g12, slab by slab as the RCCL exchange
    does it); the finish phase must reproduce the single pass bit for bit.  World = 1, 2, 3, 8
    emulated in one process."""
    import torch

    D = importlib.import_module("21cmfast_amd.distributed")
    # ts (round 5): with the x_e grid of a spin-temperature run the shard phases ride the fused loop too
    # (the filtered x_e only enters the barrier; what a rank leaves is still 5 bytes per cell)
    # cell_recomb = 0 (round 5): the filtered N_rec is an input every rank holds; with an x_e grid as well it
    # is the four-spectrum kernel (512-point z-lines)
    spec = recomb_spec(n, model=model, cell_recomb=cell_recomb, r_bubble_max=20.0, ts=int(ts))
    assert api.shard_rc_supported(spec)
    d = {k: torch.from_numpy(v).cuda() for k, v in inputs((n, n, n), seed=31, ts=ts).items()}
    if model == 1:
        d["prev_nrec"] = torch.full((1, 1, 1), 0.21, dtype=torch.float32, device="cuda")
    kw = dict(n_ion=d["n_ion"], whalo_sfr=d["whalo_sfr"], prev_nrec=d["prev_nrec"],
              prev_z_reion=d["prev_z_reion"])
    if ts:
        kw.update(xe=d["xe"], Tneutral=d["Tneutral"])
    buf0, _, rep0 = api.ionize_grids(spec, d["density"], **kw)
    assert api.ionize_last_loop_flags() & 2
    torch.cuda.synchronize()
    names = ("neutral_fraction", "z_reion", "kinetic_temperature", "ionisation_rate_G12",
             "mean_free_path", "cumulative_recombinations")
    ref = {k: getattr(buf0, k).clone() for k in names}
    assert 0.03 < float((ref["mean_free_path"] > 0).float().mean()) < 0.97
    ntot = n**3
    for world in ((1, 2, 3, 8) if n == 256 else (1, 3)):
        masks, vals = [], []
        for rank in range(world):
            fc = torch.empty(ntot, dtype=torch.uint8, device="cuda")
            g = torch.empty(ntot, dtype=torch.float32, device="cuda")
            api.ionize_shard_radii_rc(spec, rank, world, fc, g, d["density"], **kw)
            masks.append(fc)
            vals.append(g)
        # every radius index > 0 is owned by exactly one rank
        if world > 1:
            stacked = torch.stack(masks)
            top = stacked.max(0).values
            assert int(((stacked == top) & (top > 0)).sum(0).max()) == 1
        owner = D.owner_rank(spec.n_radii, world)
        fc, g = masks[owner].clone(), vals[owner].clone()
        for r in range(world):  # the combined slab of rank r, as hop 1 of the exchange leaves it
            lo, hi = D.ts_slab(ntot, r, world)
            if hi == lo:
                continue
            m, v = masks[r][lo:hi].clone(), vals[r][lo:hi].clone()
            peers = [q for q in range(world) if q != r]
            if peers:
                stride = (hi - lo + 15) // 16 * 16
                pm = torch.zeros((len(peers), stride), dtype=torch.uint8, device="cuda")
                pg = torch.zeros((len(peers), stride), dtype=torch.float32, device="cuda")
                for i, q in enumerate(peers):
                    pm[i, : hi - lo], pg[i, : hi - lo] = masks[q][lo:hi], vals[q][lo:hi]
                api.combine_cross_g12(m, v, pm, pg)
            fc[lo:hi], g[lo:hi] = m, v
        want = torch.stack(masks).max(0).values
        assert torch.equal(fc, want)
        buf, _, rep = api.ionize_shard_finish_rc(spec, fc, g, d["density"], **kw)
        torch.cuda.synchronize()
        for name in names:
            assert torch.equal(ref[name], getattr(buf, name)), (world, name)
        assert rep.global_xH == rep0.global_xH


def test_c_level_sharding_of_the_fused_recombination_loop_on_one_rank(api, monkeypatch):
    """c21cm_ionize_sharded on a one-rank RCCL communicator takes the (first crossing, Gamma_12)
    route for a spec the fused recombination loop supports, the key reduce with
    C21CM_SHARD_EXCHANGE=keys: both equal the single pass (the keys through the unfused sequence:
    same crossings up to float round-off of the barrier)."""
    import torch

    n = 256
    spec = recomb_spec(n, model=2, cell_recomb=1, r_bubble_max=20.0)
    d = {k: torch.from_numpy(v).cuda() for k, v in inputs((n, n, n), seed=9).items()}
    kw = dict(n_ion=d["n_ion"], whalo_sfr=d["whalo_sfr"], prev_nrec=d["prev_nrec"],
              prev_z_reion=d["prev_z_reion"])
    buf0, _, rep0 = api.ionize_grids(spec, d["density"], **kw)
    names = ["neutral_fraction", "z_reion", "kinetic_temperature", "ionisation_rate_G12",
             "mean_free_path", "cumulative_recombinations"]
    ref = {k: getattr(buf0, k).clone() for k in names}
    api.shard_init_single()
    try:
        buf1, _, rep1 = api.ionize_sharded(spec, d["density"], broadcast=True, **kw)
        torch.cuda.synchronize()
        for name in names:
            assert torch.equal(ref[name], getattr(buf1, name)), name
        assert rep1.global_xH == rep0.global_xH
        monkeypatch.setenv("C21CM_SHARD_EXCHANGE", "keys")
        buf2, _, rep2 = api.ionize_sharded(spec, d["density"], broadcast=True, **kw)
        torch.cuda.synchronize()
        flipped = float(((ref["neutral_fraction"] == 0) != (buf2.neutral_fraction == 0)).float().mean())
        assert flipped < 2e-4
        assert abs(rep2.global_xH - rep0.global_xH) < 2e-4
    finally:
        api.shard_finalize()
