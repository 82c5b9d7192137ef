"""Spin-temperature cell algorithm with the molecularly cooled population (USE_MINI_HALOS,
E-INTEGRAL) on the MI355X against the oracle: 2-D SFRD tables over the filtered Lyman-Werner
turnover grids, both populations in the shell loop, the J_21_LW output, the turnover-mass grid.
Tolerances as tests/test_gpu_ts.py; J_21_LW (a sum of float-rounded terms in fp64) at 2e-6.
Reference behaviour: SpinTemperatureBox.c:535-565,1011-1075,1642-1733,1843-1845."""
import importlib
import math

import numpy as np
import pytest

import ts_helpers as H
from test_gpu_ts import api, compare, to_device, to_host  # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu
S = importlib.import_module("21cmfast_amd.structs")


def run_both(api, oracle, spec, d, device):
    ref = oracle.ts_grids(spec, d["density"], d["previous"], d["source"], d["filtered_density"])
    host_mcrit = spec._keep["mcrit"]
    if device:  # device-resident turnover grids and tables
        import torch
        spec._keep["mcrit_dev"] = torch.from_numpy(host_mcrit).cuda()
        spec._keep["tabs_mini_dev"] = torch.from_numpy(spec._keep["tabs_mini"]).cuda()
        import ctypes as C
        spec.filtered_log10_mcrit = C.cast(spec._keep["mcrit_dev"].data_ptr(), S.c_float_p)
        spec.ln_sfrd_tables_mini = C.cast(spec._keep["tabs_mini_dev"].data_ptr(), S.c_float_p)
    got = api.ts_grids(spec, to_device(d["density"], device), to_device(d["previous"], device),
                       None, to_device(d["filtered_density"], device))
    if device:
        import torch
        torch.cuda.synchronize()
        spec.filtered_log10_mcrit = host_mcrit.ctypes.data_as(S.c_float_p)
        spec.ln_sfrd_tables_mini = spec._keep["tabs_mini"].ctypes.data_as(S.c_float_p)
    return got, ref


def compare_mini(got, ref, spec):
    compare(got, ref, spec)
    np.testing.assert_allclose(to_host(got["J_21_LW"]), ref["J_21_LW"], rtol=2e-6, atol=1e-30)
    n = spec.n_step
    np.testing.assert_allclose(np.array(got["report"].ave_sfrd_mini[:n]),
                               np.array(ref["report"].ave_sfrd_mini[:n]), rtol=1e-9)


@pytest.mark.parametrize("device", [False, True])
@pytest.mark.parametrize("n,n_step,nz,lya", [(24, 12, None, True), (20, 40, 36, False),
                                             (17, 9, None, True)])
def test_two_population_parity(api, oracle, device, n, n_step, nz, lya):
    spec, d = H.make(n=n, n_step=n_step, lagrangian=False, hii_dim_z=nz, lya_heating=lya)
    H.add_minis(spec, d)
    got, ref = run_both(api, oracle, spec, d, device)
    compare_mini(got, ref, spec)
    assert to_host(got["J_21_LW"]).mean() > 1e-3


def test_dark_mini_population_matches_the_tuned_one_population_kernel(api, oracle):
    spec0, d0 = H.make(n=24, n_step=12, lagrangian=False)
    base = api.ts_grids(spec0, d0["density"], d0["previous"], None, d0["filtered_density"])
    spec, d = H.make(n=24, n_step=12, lagrangian=False)
    H.add_minis(spec, d, strength=0.0)
    got, ref = run_both(api, oracle, spec, d, False)
    compare_mini(got, ref, spec)
    for k in ("kinetic_temp_neutral", "xray_ionised_fraction"):
        np.testing.assert_allclose(got[k], base[k], rtol=1e-6)


def test_no_light(api, oracle):
    spec, d = H.make(n=16, n_step=8, lagrangian=False, no_light=True)
    H.add_minis(spec, d)
    got, ref = run_both(api, oracle, spec, d, False)
    compare(got, ref, spec)
    assert (got["J_21_LW"] == 0).all()


def test_mcrit_grid_parity(api, oracle):
    shape = (12, 12, 20)
    rng = np.random.default_rng(3)
    ms = S.MturnSpec(hii_dim=12, hii_dim_z=20, redshift=15.0, vcb_const=20.0, A_LW=2.0, BETA_LW=0.6,
                     A_VCB=1.0, BETA_VCB=1.8, sigma_vcb=25.86 * math.sqrt(3 * math.pi / 8))
    j21 = (2.0 * rng.random(shape) ** 3).astype(np.float32)
    vcb = (40 * rng.random(shape)).astype(np.float32)
    for v in (vcb, None):
        for m_turn in (1e5, 10 ** 6.3):
            got = api.ts_mcrit_grid(ms, m_turn, j21, v)
            np.testing.assert_allclose(got, oracle.ts_mcrit_grid(ms, m_turn, j21, v), rtol=3e-7)
    import torch
    got = api.ts_mcrit_grid(ms, 1e5, torch.from_numpy(j21).cuda(), torch.from_numpy(vcb).cuda())
    np.testing.assert_allclose(got.cpu().numpy(), oracle.ts_mcrit_grid(ms, 1e5, j21, vcb), rtol=3e-7)


def test_refusals(api):
    spec, d = H.make(n=12, n_step=6, lagrangian=True)
    spec.use_mini_halos = 1
    with pytest.raises(RuntimeError, match="E-INTEGRAL"):
        api.ts_grids(spec, d["density"], d["previous"], d["source"], None)
