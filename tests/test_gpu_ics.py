"""GPU parity: ComputeInitialConditions grid algorithm vs the CPU oracle.

Both sides share the Philox + Box-Muller mode sampler (the reference's GSL streams are
unpinned, see oracle/oracle_ics.c), so sampled fields agree to float round-off; the
`density_is_input` path is RNG-free.  Tolerance: atol = 3e-5 * max|field| (15 chained
float32 FFTs and sweeps), the reference's own round-trip test uses atol 1e-5 on O(1) fields.
"""

import importlib

import numpy as np
import pytest

from test_oracle_ics import LOWRES_FIELDS, ics_spec

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def api(gpu_lib):
    return importlib.import_module("21cmfast_amd.grid_api")


def compare(got, ref, names=None):
    for k in names or ref.keys():
        a = got[k] if isinstance(got[k], np.ndarray) else got[k].cpu().numpy()
        scale = np.abs(ref[k]).max()
        np.testing.assert_allclose(a, ref[k], atol=3e-5 * scale + 1e-12, rtol=1e-4, err_msg=k)


@pytest.mark.parametrize("dim,hii_dim,device", [(24, 8, None), (32, 16, "cuda"), (64, 32, "cuda"),
                                                (128, 64, "cuda"), (20, 10, None)])
def test_sampled_ics_match_oracle(api, oracle, dim, hii_dim, device):
    spec = ics_spec(dim, hii_dim, box_len=3.0 * hii_dim, seed=99)
    ref = oracle.ics_grids(spec)
    got = api.ics_grids(spec, device=device)
    compare(got, ref)


@pytest.mark.parametrize("opts", [dict(algorithm=1), dict(hires=1), dict(hires=1, algorithm=1)])
def test_option_branches(api, oracle, opts):
    spec = ics_spec(32, 16, box_len=48.0, seed=5, **opts)
    compare(api.ics_grids(spec), oracle.ics_grids(spec))


@pytest.mark.parametrize("dim,hii_dim,device,opts", [
    (256, 64, "cuda", {}),                      # fold by 4
    (128, 64, None, dict(hires=1)),             # hi-res velocities, folded low-res density
    (128, 64, "cuda", dict(algorithm=1)),       # Zel'dovich only
    (128, 64, None, dict(hires=1, algorithm=1)),
    (64, 64, "cuda", {}),                       # DIM == HII_DIM: no filter, no fold
])
def test_split_layout_pipeline_matches_oracle(api, oracle, dim, hii_dim, device, opts):
    """The split-layout pipeline (native transform sizes): spectra in the split layout, dense
    stores from pass Z, low-resolution outputs from the folded spectrum."""
    spec = ics_spec(dim, hii_dim, box_len=1.5 * dim, seed=31, **opts)
    compare(api.ics_grids(spec, device=device), oracle.ics_grids(spec))


def test_split_and_padded_pipelines_agree(api, monkeypatch):
    """C21CM_ICS=padded selects the padded-layout pipeline (full-size transforms + gathers);
    folding is an exact identity, so the two agree to transform round-off."""
    spec = ics_spec(128, 64, box_len=192.0, seed=77)
    a = api.ics_grids(spec, device="cuda")
    monkeypatch.setenv("C21CM_ICS", "padded")
    b = api.ics_grids(spec, device="cuda")
    for k in a:
        x, y = a[k].cpu().numpy(), b[k].cpu().numpy()
        np.testing.assert_allclose(x, y, atol=2e-5 * np.abs(y).max(), rtol=1e-4, err_msg=k)
    assert not all(np.array_equal(a[k].cpu().numpy(), b[k].cpu().numpy()) for k in LOWRES_FIELDS)


def test_density_input_on_the_split_pipeline(api, oracle):
    dim, hii = 128, 64
    spec = ics_spec(dim, hii, box_len=192.0, seed=8)
    ic = api.ics_grids(spec)
    spec2 = ics_spec(dim, hii, box_len=192.0, density_is_input=1)
    start = api.new_ics_arrays(spec2)
    start["hires_density"][...] = ic["hires_density"]
    ic2 = api.ics_grids(spec2, start)
    for name in LOWRES_FIELDS:
        scale = max(1.0, np.abs(ic[name]).max())
        np.testing.assert_allclose(ic[name], ic2[name], atol=1e-5 * scale, rtol=0.0, err_msg=name)
    ref = oracle.new_ics_arrays(spec2)
    ref["hires_density"][...] = ic["hires_density"]
    compare(ic2, oracle.ics_grids(spec2, ref), LOWRES_FIELDS)


def test_roundtrip_from_own_density_on_device(api, oracle):
    """Reference test tests/test_initial_conditions.py:153-167 on the HIP path."""
    dim, hii = 64, 32
    spec = ics_spec(dim, hii, box_len=96.0, seed=7)
    ic = api.ics_grids(spec)
    spec2 = ics_spec(dim, hii, box_len=96.0, density_is_input=1)
    start = api.new_ics_arrays(spec2)
    start["hires_density"][...] = ic["hires_density"]
    ic2 = api.ics_grids(spec2, start)
    assert np.all(ic2["hires_density"] == ic["hires_density"])
    for name in LOWRES_FIELDS:
        scale = max(1.0, np.abs(ic[name]).max())
        np.testing.assert_allclose(ic[name], ic2[name], atol=1e-5 * scale, rtol=0.0, err_msg=name)
    # and the input path agrees with the oracle run on the same density
    ref = oracle.new_ics_arrays(spec2)
    ref["hires_density"][...] = ic["hires_density"]
    compare(ic2, oracle.ics_grids(spec2, ref), LOWRES_FIELDS)


def test_full_size_properties(api):
    """Config 2 (HII_DIM=256, DIM=512): zero-mean fields, realisation independent of launch
    geometry (same seed twice -> identical bits), different seed -> different field."""
    import torch

    spec = ics_spec(512, 256, box_len=384.0, seed=12345)
    a = api.ics_grids(spec, device="cuda")
    b = api.ics_grids(spec, device="cuda")
    torch.cuda.synchronize()
    for k in a:
        assert torch.equal(a[k], b[k]), k
        assert abs(a[k].double().mean().item()) < 1e-5 * max(1.0, a[k].abs().max().item()), k
    c = api.ics_grids(ics_spec(512, 256, box_len=384.0, seed=54321), device="cuda")
    assert not torch.equal(a["hires_density"], c["hires_density"])
    # sigma of the low-res density is below the hi-res one (top-hat smoothing)
    assert a["lowres_density"].std() < a["hires_density"].std()
