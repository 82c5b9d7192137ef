"""Per-cell part of ComputeTsBox on the MI355X against the CPU oracle (reference:
src/py21cmfast/src/SpinTemperatureBox.c:892-927 init_first_Ts, :1010-1086 SFRD from the filtered
density, :1210-1383 get_Ts_fast, :1499-1848 the R loop and the cell outputs).

Tolerances: everything is evaluated in double from float inputs and rounded to float at the end;
the device's exp / log / pow / cbrt differ from glibc's in the last bits, and the T_s fixed point
stops at a 1e-3 relative step, so two evaluations that differ in the last bit can stop one
iteration apart: T_s is compared at 2e-3, x_e and T_k at 2e-6."""

import importlib

import numpy as np
import pytest

import ts_helpers as H

pytestmark = pytest.mark.gpu
S = importlib.import_module("21cmfast_amd.structs")


@pytest.fixture(scope="module")
def api(gpu_lib):
    return importlib.import_module("21cmfast_amd.grid_api")


def to_device(x, device):
    import torch

    if not device or x is None:
        return x
    if isinstance(x, dict):
        return {k: to_device(v, device) for k, v in x.items()}
    return torch.from_numpy(x).cuda()


def to_host(x):
    return x.cpu().numpy() if hasattr(x, "cpu") else x


def run_both(api, oracle, spec, d, device):
    ref = oracle.ts_grids(spec, d["density"], d["previous"], d["source"], d["filtered_density"])
    got = api.ts_grids(spec, to_device(d["density"], device), to_device(d["previous"], device),
                       to_device(d["source"], device), to_device(d["filtered_density"], device))
    return got, ref


def compare(got, ref, spec):
    for k, tol in (("xray_ionised_fraction", 2e-6), ("kinetic_temp_neutral", 2e-6),
                   ("spin_temperature", 2e-3)):
        np.testing.assert_allclose(to_host(got[k]), ref[k], rtol=tol, atol=1e-30, err_msg=k)
    # most cells stop the T_s iteration at the same step: then they agree like the others
    rel = np.abs(to_host(got["spin_temperature"]) / ref["spin_temperature"] - 1)
    assert np.mean(rel < 5e-6) > 0.99
    for f in ("Ts_ave", "Tk_ave", "x_e_ave", "J_alpha_ave", "xheat_ave", "xion_ave"):
        assert getattr(got["report"], f) == pytest.approx(getattr(ref["report"], f), rel=1e-5), f


@pytest.mark.parametrize("device", [False, True])
@pytest.mark.parametrize("lagrangian,n,n_step,nz", [
    (True, 24, 12, None), (False, 24, 12, None),
    (True, 20, 40, 28),   # the default N_STEP_TS, non-cubic box
    (False, 16, 40, None),
])
def test_cell_sweep_matches_oracle(api, oracle, lagrangian, n, n_step, nz, device):
    spec, d = H.make(n=n, n_step=n_step, lagrangian=lagrangian, hii_dim_z=nz)
    got, ref = run_both(api, oracle, spec, d, device)
    compare(got, ref, spec)
    if not lagrangian:
        np.testing.assert_allclose(np.array(got["report"].ave_sfrd[:n_step]),
                                   np.array(ref["report"].ave_sfrd[:n_step]), rtol=1e-9)
    # the special cells of the workload (ts_helpers.make) went through their branches
    Ts = to_host(got["spin_temperature"])
    assert np.isfinite(Ts).all()
    assert to_host(got["kinetic_temp_neutral"]).flat[13] == np.float32(6e4)  # frozen above MAX_TK


@pytest.mark.parametrize("flags", [
    dict(lya_heating=False), dict(xray_heating=False), dict(cmb_heating=False),
    dict(no_light=True), dict(lya_heating=False, xray_heating=False, cmb_heating=False),
])
def test_option_switches(api, oracle, flags):
    for lagrangian in (True, False):
        spec, d = H.make(n=16, n_step=8, lagrangian=lagrangian, **flags)
        got, ref = run_both(api, oracle, spec, d, False)
        compare(got, ref, spec)


def test_first_box(api, oracle):
    """init_first_Ts: redshift >= Z_HEAT_MAX."""
    fs = H.first_spec(n=20, hii_dim_z=24)
    dens = H.smooth_field((20, 20, 24), np.random.default_rng(2), 0.06)
    ref = oracle.ts_first_grids(fs, dens)
    for device in (False, True):
        got = api.ts_first_grids(fs, to_device(dens, device))
        for k in ref:
            np.testing.assert_allclose(to_host(got[k]), ref[k], rtol=2e-6, err_msg=k)


def test_requests_are_validated(api):
    BackendError = importlib.import_module("21cmfast_amd").BackendError
    spec, d = H.make(n=8, n_step=4)
    with pytest.raises(BackendError, match="filtered_sfr"):
        api.ts_grids(spec, d["density"], d["previous"], None, None)
    spec.lya_dEC = None
    with pytest.raises(BackendError, match="USE_LYA_HEATING"):
        api.ts_grids(spec, d["density"], d["previous"], d["source"], None)
    spec, d = H.make(n=8, n_step=4, lagrangian=False)
    with pytest.raises(BackendError, match="filtered densities"):
        api.ts_grids(spec, d["density"], d["previous"], None, None)
    spec.n_step = 500
    with pytest.raises(BackendError, match="shells"):
        api.ts_grids(spec, d["density"], d["previous"], None, d["filtered_density"])


def test_nan_inputs_are_reported(api):
    """A non-finite spin temperature is an InfinityorNaNError upstream (:1884-1904)."""
    BackendError = importlib.import_module("21cmfast_amd").BackendError
    spec, d = H.make(n=8, n_step=4)
    d["previous"]["kinetic_temp_neutral"].flat[40] = np.nan
    with pytest.raises(BackendError):
        api.ts_grids(spec, d["density"], d["previous"], d["source"], None)
