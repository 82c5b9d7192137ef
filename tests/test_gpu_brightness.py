"""GPU: ComputeBrightnessTemp on the MI355X vs the CPU oracle (bit-exact without spin
temperatures: the sweep is three float operations; rtol 1e-6 through exp() with them), through
the grid API on host and device arrays and through the reference's entry point."""

import ctypes as C
import importlib

import numpy as np
import pytest

from test_oracle_brightness import fields

pytestmark = pytest.mark.gpu
S = importlib.import_module("21cmfast_amd.structs")


@pytest.fixture(scope="module")
def api(gpu_lib):
    return importlib.import_module("21cmfast_amd.grid_api")


@pytest.mark.parametrize("use_ts", [False, True])
@pytest.mark.parametrize("device", [False, True])
def test_brightness_matches_oracle(api, oracle, use_ts, device):
    density, xH, Ts = fields(n=40, seed=11)
    spec = S.brightness_spec(density.size, 7.6, use_ts_fluct=use_ts)
    ref = oracle.brightness_grids(spec, density, xH, Ts if use_ts else None)
    if device:
        import torch

        args = [torch.from_numpy(a).cuda() for a in (density, xH, Ts)]
    else:
        args = [density, xH, Ts]
    got = api.brightness_grids(spec, args[0], args[1], args[2] if use_ts else None)
    bt = got["brightness_temp"].cpu().numpy() if device else got["brightness_temp"]
    if use_ts:
        tau = got["tau_21"].cpu().numpy() if device else got["tau_21"]
        np.testing.assert_array_equal(tau, ref["tau_21"])
        np.testing.assert_allclose(bt, ref["brightness_temp"], rtol=1e-6, atol=1e-7)
    else:
        np.testing.assert_array_equal(bt, ref["brightness_temp"])
    assert got["mean"] == pytest.approx(ref["mean"], rel=1e-9)


def test_entry_point(gpu_lib, oracle, tmp_path):
    from test_gpu_abi import Session, fptr

    ses = Session(gpu_lib, tmp_path, HII_DIM=24)  # noqa: F841  (keeps the parameter structs alive)
    density, xH, _ = fields(n=24, seed=2)
    bt = np.zeros_like(density)
    pf = S.PerturbedFieldStruct(density=fptr(density))
    ion = S.IonizedBoxStruct(neutral_fraction=fptr(xH))
    box = S.BrightnessTempStruct(brightness_temp=fptr(bt))
    ts = S.TsBoxStruct()
    gpu_lib.ComputeBrightnessTemp.argtypes = [C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    st = gpu_lib.ComputeBrightnessTemp(8.0, C.byref(ts), C.byref(ion), C.byref(pf), C.byref(box))
    assert st == 0, gpu_lib.c21cm_last_error()
    ref = oracle.brightness_grids(S.brightness_spec(density.size, 8.0, cosmo=ses.cp), density, xH)
    np.testing.assert_array_equal(bt, ref["brightness_temp"])
    # a non-finite input is reported with the reference status code (InfinityorNaNError = 7)
    xH[0, 0, 0] = np.inf
    assert gpu_lib.ComputeBrightnessTemp(8.0, C.byref(ts), C.byref(ion), C.byref(pf),
                                         C.byref(box)) == 7
