"""GPU parity: ComputePerturbedField grid algorithm (CIC deposit with fp64 atomics, FFTs,
clip, velocity) vs the CPU oracle, and the reference's own known-answer test on the device.

Tolerance: density and velocity are float32 results of two FFTs on O(1) data; the fp64
atomic deposit is order-dependent at the 1e-16 level only.  atol = 2e-5 * max|field|.
"""

import importlib

import numpy as np
import pytest

from test_oracle_perturb import expected_density, fake_ics, perturb_spec

pytestmark = pytest.mark.gpu
S = importlib.import_module("21cmfast_amd.structs")


@pytest.fixture(scope="module")
def api(gpu_lib):
    return importlib.import_module("21cmfast_amd.grid_api")


@pytest.mark.parametrize("algorithm", [2, 1, 0])
def test_reference_known_answer_on_device(api, algorithm):
    """tests/test_perturb.py:108-135 of the reference, through the HIP path."""
    ics = fake_ics(algorithm)
    out = api.perturb_grids(perturb_spec(algorithm), ics)
    np.testing.assert_allclose(out["density"], expected_density(ics, algorithm), atol=1e-3)


def random_ics(n, N, seed, hires_vel=False):
    rng = np.random.default_rng(seed)
    vshape = (N,) * 3 if hires_vel else (n,) * 3
    pre = "hires" if hires_vel else "lowres"
    ics = {}
    for ax in "xyz":
        ics[f"{pre}_v{ax}"] = (1.5 * rng.standard_normal(vshape)).astype(np.float32)
        ics[f"{pre}_v{ax}_2LPT"] = (0.8 * rng.standard_normal(vshape)).astype(np.float32)
    d = (2.0 * rng.standard_normal((N,) * 3)).astype(np.float32)
    ics["hires_density"] = d - d.mean()
    ics["lowres_density"] = (0.3 * rng.standard_normal((n,) * 3)).astype(np.float32)
    return ics


def compare(got, ref):
    for k in ref:
        scale = np.abs(ref[k]).max()
        np.testing.assert_allclose(got[k], ref[k], atol=2e-5 * scale + 1e-9, rtol=1e-4,
                                   err_msg=k)


@pytest.mark.parametrize("n,N,device", [(16, 32, False), (32, 64, True), (64, 128, True),
                                        (25, 50, False), (12, 36, True)])
@pytest.mark.parametrize("algorithm", [2, 1])
def test_lowres_perturb_matches_oracle(api, oracle, n, N, device, algorithm):
    ics = random_ics(n, N, seed=n + algorithm)
    spec = perturb_spec(algorithm, dim=N, dim_z=N, hii_dim=n, hii_dim_z=n, box_len=1.5 * n,
                        box_len_z=1.5 * n, growth_factor=0.12, init_growth_factor=0.0042,
                        keep_3d_velocities=1, dDdt_over_D=2.1e-17)
    ref = oracle.perturb_grids(spec, ics)
    if device:
        import torch

        ics_d = {k: torch.from_numpy(v).cuda() for k, v in ics.items()}
        got = {k: v.cpu().numpy() for k, v in api.perturb_grids(spec, ics_d).items()}
    else:
        got = api.perturb_grids(spec, ics)
    compare(got, ref)


@pytest.mark.parametrize("opts", [dict(perturb_on_high_res=1), dict(smooth_evolved_density=1),
                                  dict(perturb_on_high_res=1, perturb_algorithm=0),
                                  dict(perturb_algorithm=0)])
def test_option_branches(api, oracle, opts):
    n, N = 16, 32
    hires = bool(opts.get("perturb_on_high_res"))
    ics = random_ics(n, N, seed=3, hires_vel=hires)
    kw = dict(dim=N, dim_z=N, hii_dim=n, hii_dim_z=n, box_len=24.0, box_len_z=24.0,
              growth_factor=0.1, init_growth_factor=0.004, dDdt_over_D=2e-17,
              density_smooth_radius_mpc=0.8 * 24.0 / n)
    kw.update(opts)
    algorithm = kw.pop("perturb_algorithm", 2)
    spec = perturb_spec(algorithm, **kw)
    compare(api.perturb_grids(spec, ics), oracle.perturb_grids(spec, ics))


@pytest.mark.parametrize("opts", [dict(smooth_evolved_density=1), dict(keep_3d_velocities=0),
                                  dict(perturb_algorithm=1, smooth_evolved_density=1)])
def test_split_layout_kspace_part(api, oracle, opts, monkeypatch):
    """Low-resolution branch at a size the native transform covers (64^3 from 128^3): the
    k-space part runs on the split layout (smoothing window as one sweep, velocity operator in
    pass X, dense stores with / N and the density floor from pass Z); C21CM_PT=padded selects
    the padded-layout sequence, and both agree with the oracle."""
    n, N = 64, 128
    ics = random_ics(n, N, seed=11)
    kw = dict(dim=N, dim_z=N, hii_dim=n, hii_dim_z=n, box_len=96.0, box_len_z=96.0,
              growth_factor=0.11, init_growth_factor=0.004, dDdt_over_D=2e-17,
              keep_3d_velocities=1, density_smooth_radius_mpc=0.9 * 96.0 / n)
    kw.update(opts)
    algorithm = kw.pop("perturb_algorithm", 2)
    spec = perturb_spec(algorithm, **kw)
    ref = oracle.perturb_grids(spec, ics)
    got = api.perturb_grids(spec, ics)
    compare(got, ref)
    monkeypatch.setenv("C21CM_PT", "padded")
    compare(api.perturb_grids(spec, ics), ref)
    assert got["density"].min() >= np.float32(-1.0 + 1e-7)


def test_full_size_mass_conservation(api):
    """Config 2 (HII_DIM=256, DIM=512): size-independent properties.
    mean(delta) = 0 (mass conservation of the CIC deposit), delta >= -1, zero-mean velocity."""
    import torch

    n, N = 256, 512
    g = torch.Generator(device="cuda").manual_seed(1)
    ics = {}
    for ax in "xyz":
        ics[f"lowres_v{ax}"] = 2.0 * torch.randn((n,) * 3, generator=g, device="cuda")
        ics[f"lowres_v{ax}_2LPT"] = 1.0 * torch.randn((n,) * 3, generator=g, device="cuda")
    d = torch.randn((N,) * 3, generator=g, device="cuda")
    ics["hires_density"] = (d - d.mean()).contiguous()
    spec = perturb_spec(2, dim=N, dim_z=N, hii_dim=n, hii_dim_z=n, box_len=384.0, box_len_z=384.0,
                        growth_factor=0.127, init_growth_factor=0.0042, dDdt_over_D=2e-17)
    out = api.perturb_grids(spec, ics)
    torch.cuda.synchronize()
    dens = out["density"].double()
    assert abs(dens.mean().item()) < 1e-5
    assert dens.min().item() >= -1.0
    assert abs(out["velocity_z"].double().mean().item()) < 1e-9


@pytest.mark.parametrize("n,N,vscale", [(256, 512, 2.0), (512, 1024, 2.0), (128, 256, 40.0), (128, 384, 3.0)])
def test_deposit_is_deterministic(api, n, N, vscale, monkeypatch):
    """Round 5: the deposit accumulates 64-bit fixed-point integers (scale 2^44) in LDS and in the grid --
    integer additions commute, so repeated calls give the SAME BITS whatever order the hardware serves
    the atomics in (the fp64 atomics of rounds 1-4 flipped last bits between calls; upstream's OpenMP
    atomics are unordered too, map_mass.c:197-206).  The large-displacement case sends cells through the
    queued global path as well.  Against the fp64 accumulation the densities agree to a float ulp."""
    import torch

    g = torch.Generator(device="cuda").manual_seed(7)
    ics = {}
    for ax in "xyz":
        ics[f"lowres_v{ax}"] = vscale * torch.randn((n,) * 3, generator=g, device="cuda")
        ics[f"lowres_v{ax}_2LPT"] = 0.5 * vscale * torch.randn((n,) * 3, generator=g, device="cuda")
    d = torch.randn((N,) * 3, generator=g, device="cuda")
    ics["hires_density"] = (d - d.mean()).contiguous()
    spec = perturb_spec(2, dim=N, dim_z=N, hii_dim=n, hii_dim_z=n, box_len=1.5 * n, box_len_z=1.5 * n,
                        growth_factor=0.127, init_growth_factor=0.0042, dDdt_over_D=2e-17)
    first = api.perturb_grids(spec, ics)["density"].clone()
    for _ in range(4):
        again = api.perturb_grids(spec, ics)["density"]
        torch.cuda.synchronize()
        assert torch.equal(first, again)
    monkeypatch.setenv("C21CM_CIC_ACC", "double")
    dbl = api.perturb_grids(spec, ics)["density"]
    torch.cuda.synchronize()
    scale = float((1 + first).abs().max())
    assert float((first - dbl).abs().max()) <= 2.5e-7 * scale
    assert abs(first.double().mean().item()) < 1e-5


def test_non_finite_density_is_an_error_not_a_finite_wrong_field(api):
    """the fixed-point integers of the deposit wrap silently -- a NaN / Inf hi-res density used to come
    back as a finite, wrong density grid.  The deposit flags it and the call returns the reference's
    InfinityorNaNError status (7)."""
    import torch

    n, N = 64, 128
    g = torch.Generator(device="cuda").manual_seed(3)
    for poison in (float("nan"), float("inf")):
        ics = {}
        for ax in "xyz":
            ics[f"lowres_v{ax}"] = torch.randn((n,) * 3, generator=g, device="cuda")
            ics[f"lowres_v{ax}_2LPT"] = 0.5 * torch.randn((n,) * 3, generator=g, device="cuda")
        d = torch.randn((N,) * 3, generator=g, device="cuda")
        d[17, 40, 99] = poison
        ics["hires_density"] = d.contiguous()
        spec = perturb_spec(2, dim=N, dim_z=N, hii_dim=n, hii_dim_z=n, box_len=1.5 * n, box_len_z=1.5 * n,
                            growth_factor=0.127, init_growth_factor=0.0042, dDdt_over_D=2e-17)
        with pytest.raises(RuntimeError, match="status 7"):
            api.perturb_grids(spec, ics)
    # and the library is fine afterwards
    d = torch.randn((N,) * 3, generator=g, device="cuda")
    ics["hires_density"] = (d - d.mean()).contiguous()
    out = api.perturb_grids(spec, ics)["density"]
    assert bool(torch.isfinite(out).all())


@pytest.mark.parametrize("vscale", [1.0, 12.0, 60.0])
def test_deposit_paths_large_displacements(api, oracle, vscale):
    """The LDS-tiled deposit keeps particles that leave the tile halo (2 output cells) on a
    direct global-atomic path: scale the displacements from well inside the halo to several
    cells beyond it (and around the periodic box) and compare with the oracle."""
    n, N = 32, 64
    ics = random_ics(n, N, seed=5)
    for k in list(ics):
        if k.startswith("lowres_v"):
            ics[k] = (ics[k] * vscale).astype(np.float32)
    spec = perturb_spec(2, dim=N, dim_z=N, hii_dim=n, hii_dim_z=n, box_len=1.5 * n,
                        box_len_z=1.5 * n, growth_factor=0.12, init_growth_factor=0.0042,
                        keep_3d_velocities=0, dDdt_over_D=2.1e-17)
    ref = oracle.perturb_grids(spec, ics)
    got = api.perturb_grids(spec, ics)
    compare(got, ref)
    assert ref["density"].std() > 0


@pytest.mark.parametrize("n,N,nz,hires", [(48, 96, 64, False), (32, 96, 32, False),
                                          (32, 128, 48, False), (40, 40, 40, True)])
def test_deposit_implementations_agree(api, n, N, nz, hires, monkeypatch):
    """The three deposits -- per velocity cell with 27 merged LDS atomics (default; DIM / HII_DIM
    = 1 ... 4), per particle into the LDS tile (rounds 1-3) and plain global atomics -- sum the same
    doubles in different orders: the float densities agree to rounding, also on a non-cubic box,
    with the first velocity plane's sources wrapping around the box and with a tail of particles
    far beyond the tile halo."""
    f = N // n
    Nz = nz * f
    rng = np.random.default_rng(n + N)
    vshape = (N, N, Nz) if hires else (n, n, nz)
    pre = "hires" if hires else "lowres"
    ics = {}
    for ax in "xyz":
        v = 1.5 * rng.standard_normal(vshape)
        v[rng.random(vshape) < 0.02] *= 8.0  # a few far-flung cells: the queued global path
        ics[f"{pre}_v{ax}"] = v.astype(np.float32)
        ics[f"{pre}_v{ax}_2LPT"] = (0.8 * rng.standard_normal(vshape)).astype(np.float32)
    d = (2.0 * rng.standard_normal((N, N, Nz))).astype(np.float32)
    ics["hires_density"] = d - d.mean()
    spec = perturb_spec(2, dim=N, dim_z=Nz, hii_dim=n, hii_dim_z=nz, box_len=1.5 * n,
                        box_len_z=1.5 * nz, growth_factor=0.12, init_growth_factor=0.0042,
                        keep_3d_velocities=0, dDdt_over_D=2.1e-17,
                        perturb_on_high_res=1 if hires else 0)
    out = {}
    for mode in ("cell", "tiled", "direct"):
        monkeypatch.setenv("C21CM_CIC", mode)
        out[mode] = api.perturb_grids(spec, ics)["density"]
    scale = np.abs(out["direct"]).max()
    assert out["direct"].std() > 0
    for mode in ("cell", "tiled"):
        np.testing.assert_allclose(out[mode], out["direct"], atol=3e-7 * scale, rtol=2e-6,
                                   err_msg=mode)
