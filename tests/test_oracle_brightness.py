"""CPU: the oracle's ComputeBrightnessTemp sweep against a direct numpy evaluation of
BrightnessTemperatureBox.c:43-87 (float products left to right, the spin-temperature branch in
double), including the saturated-spin-temperature limit delta_T = 27 x_HI (1+delta) sqrt(...) mK."""

import importlib

import numpy as np
import pytest

S = importlib.import_module("21cmfast_amd.structs")


def fields(n=24, seed=3):
    rng = np.random.default_rng(seed)
    density = (0.4 * rng.standard_normal((n, n, n))).astype(np.float32).clip(-0.9, None)
    xH = rng.uniform(0, 1, (n, n, n)).astype(np.float32)
    xH[rng.uniform(size=xH.shape) < 0.3] = 0.0
    Ts = rng.uniform(5.0, 400.0, (n, n, n)).astype(np.float32)
    return density, xH, Ts


def numpy_brightness(spec, density, xH, Ts=None):
    cf, trad, z = np.float32(spec.const_factor), np.float32(spec.T_rad), spec.redshift
    bt = (cf * xH) * (np.float32(1) + density)
    assert bt.dtype == np.float32
    if not spec.use_ts_fluct:
        return bt, None
    tau = (bt.astype(np.float64) * ((1.0 + z) / (1000.0 * Ts.astype(np.float64)))).astype(np.float32)
    out = (1.0 - np.exp(-tau.astype(np.float64))) * 1000.0 * (Ts - trad).astype(np.float64) / (1.0 + z)
    return out.astype(np.float32), tau


@pytest.mark.parametrize("use_ts", [False, True])
def test_oracle_brightness_matches_formula(oracle, use_ts):
    density, xH, Ts = fields()
    spec = S.brightness_spec(density.size, 8.3, use_ts_fluct=use_ts)
    out = oracle.brightness_grids(spec, density, xH, Ts if use_ts else None)
    want, tau = numpy_brightness(spec, density, xH, Ts)
    if use_ts:
        np.testing.assert_array_equal(out["tau_21"], tau)
        np.testing.assert_allclose(out["brightness_temp"], want, rtol=2e-7, atol=1e-9)
    else:
        np.testing.assert_array_equal(out["brightness_temp"], want)
    assert out["mean"] == pytest.approx(out["brightness_temp"].astype(np.float64).sum()
                                        / np.float32(density.size), rel=1e-12)


def test_const_factor_value():
    """27 (Ob h^2 / 0.023) sqrt(0.15 / (Om h^2) (1+z)/10) mK for Planck18 at z = 9."""
    spec = S.brightness_spec(1, 9.0)
    h = 0.6766  # the reference's Planck18: inputs.py:126-134
    om, ob = (0.02242 + 0.11933) / h**2, 0.02242 / h**2
    want = 27 * (ob * h * h / 0.023) * np.sqrt(0.15 / (om * h * h) * 10.0 / 10.0)
    assert spec.const_factor == pytest.approx(want, rel=1e-6)
    assert spec.T_rad == pytest.approx(27.255, rel=1e-6)
