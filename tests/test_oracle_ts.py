"""The oracle's restatement of the per-cell part of ComputeTsBox, checked against closed forms
and limits of the algorithm (reference: src/py21cmfast/src/SpinTemperatureBox.c:892-927,
1210-1383, 1499-1848; heating_helper_progs.c:366-643)."""

import math

import numpy as np
import pytest

import ts_helpers as H
from oracle import ref_heating as RH


def test_kappa_tables_follow_their_knots(oracle):
    lib = oracle.load()
    # the first knot of each table is ln T = 0, values are ln kappa (heating_helper_progs.c)
    assert lib.oracle_kappa_10(1.0) == pytest.approx(math.exp(-29.6115227098), rel=1e-12)
    assert lib.oracle_kappa_10_pH(1.0) == pytest.approx(math.exp(-21.6395565688), rel=1e-12)
    assert lib.oracle_kappa_10_elec(1.0) == pytest.approx(math.exp(-22.1549007191), rel=1e-12)
    # below 1 K the first value is used; H-H continues as T^0.381 above the table
    assert lib.oracle_kappa_10(0.3) == lib.oracle_kappa_10(1.0)
    top = math.exp(9.21034037198)
    assert lib.oracle_kappa_10(4 * top) / lib.oracle_kappa_10(top) == pytest.approx(4**0.381, rel=1e-9)
    # half-way between two knots in ln T: the geometric mean of the knot values
    w = 0.317597943861
    mid = lib.oracle_kappa_10(math.exp(2.5 * w))
    assert mid == pytest.approx(math.sqrt(math.exp(-29.5917673123) * math.exp(-29.4469989515)),
                                rel=1e-9)
    # physical magnitudes (Zygelman 2005: kappa_HH(100 K) ~ 2e-11 cm^3/s... in these units)
    assert 1e-13 < lib.oracle_kappa_10(300.0) < 1e-9


def test_alpha_A_is_the_abel_fit(oracle):
    lib = oracle.load()
    assert lib.oracle_alpha_A(1e4) == pytest.approx(4.2e-13, rel=0.05)   # Osterbrock 4.18e-13
    assert lib.oracle_alpha_A(1e3) > lib.oracle_alpha_A(1e4) > lib.oracle_alpha_A(1e5)


def test_lya_efficiency_is_trilinear_and_clamped(oracle):
    lib = oracle.load()
    dEC, _ = H.lya_tables(None)
    p = dEC.ctypes.data
    # on a node: the table value itself
    tk, ts, tg = 10.0 ** (-1 + 40 * 0.04), 10.0 ** (-1 + 55 * 0.04), 10.0 ** (1 + 20 * 0.12)
    assert lib.oracle_lya_heating_efficiency(tk, ts, tg, p) == pytest.approx(dEC[40, 55, 20], rel=1e-9)
    # outside the table: the edge value
    assert lib.oracle_lya_heating_efficiency(1e-3, 1e5, 1e9, p) == pytest.approx(dEC[0, 100, 50], rel=1e-12)
    # inside a cell: scipy's regular-grid interpolator
    from scipy.interpolate import RegularGridInterpolator

    t, g = np.linspace(-1, 3, 101), np.linspace(1, 7, 51)
    f = RegularGridInterpolator((t, t, g), dEC)
    for a, b, c in ((33.0, 71.0, 4.2e3), (2.5, 900.0, 77.0), (480.0, 12.0, 3.3e6)):
        assert lib.oracle_lya_heating_efficiency(a, b, c, p) == pytest.approx(
            f([math.log10(a), math.log10(b), math.log10(c)])[0], rel=1e-9)


def test_no_sources_is_adiabatic_compton_recombination_only(oracle):
    """no_light: the radiative sums stay zero, so x_e only recombines and T_k follows the
    closed-form derivative; T_s is the collisional / CMB equilibrium between T_k and T_cmb."""
    spec, d = H.make(n=12, n_step=6, no_light=True, lya_heating=False, cmb_heating=False)
    out = oracle.ts_grids(spec, d["density"], d["previous"], d["source"], d["filtered_density"])
    prev = d["previous"]
    xe0, Tk0 = prev["xray_ionised_fraction"].astype(float), prev["kinetic_temp_neutral"].astype(float)
    delta = np.maximum(d["density"].astype(float), -1 + 1e-7)
    lib = oracle.load()
    alpha = np.vectorize(lib.oracle_alpha_A)(Tk0)
    sink = alpha * spec.clumping_factor * xe0**2 * spec.h_frac * spec.Nb_zp * (1 + delta)
    dxe_dzp = -spec.dt_dzp * sink
    xe1 = np.clip(xe0 + dxe_dzp * spec.dzp, 0, 1)
    np.testing.assert_allclose(out["xray_ionised_fraction"], xe1, rtol=2e-6)
    assert np.all(out["xray_ionised_fraction"] <= prev["xray_ionised_fraction"])
    with np.errstate(divide="ignore"):
        adia = 3 / (1 + spec.redshift) + np.where(
            np.abs(delta) > 1e-7, spec.dgrowth_dzp / (spec.growth_zp * (1 / delta + 1)), 0.0)
    adia *= (2 / 3) * Tk0
    dspec = -dxe_dzp * Tk0 / (1 + xe0)
    dcomp = spec.dcomp_dzp_prefactor * (xe0 / (1 + xe0 + spec.he_frac)) * (spec.Trad - Tk0)
    Tk1 = np.where(Tk0 < 5e4, Tk0 + (adia + dspec + dcomp) * spec.dzp, Tk0)
    Tk1 = np.where(Tk1 < 0, spec.Trad, Tk1)
    np.testing.assert_allclose(out["kinetic_temp_neutral"], Tk1, rtol=2e-6)
    Ts, Tk = out["spin_temperature"].astype(float), out["kinetic_temp_neutral"].astype(float)
    lo, hi = np.minimum(Tk, spec.Trad), np.maximum(Tk, spec.Trad)
    assert np.all((Ts >= lo * (1 - 1e-6)) & (Ts <= hi * (1 + 1e-6)))
    assert out["report"].J_alpha_ave == 0 and out["report"].xheat_ave == 0


@pytest.mark.parametrize("lagrangian", [True, False])
def test_magnitudes_and_branches_of_the_parity_workload(oracle, lagrangian):
    spec, d = H.make(n=16, n_step=8, lagrangian=lagrangian)
    out = oracle.ts_grids(spec, d["density"], d["previous"], d["source"], d["filtered_density"])
    rep = out["report"]
    # physical magnitudes: X-rays ionise and heat, the Lyman-alpha flux couples
    assert 5e-5 < rep.xion_ave * abs(spec.dzp * spec.dt_dzp) < 5e-3
    assert 1e-11 < rep.J_alpha_ave < 1e-8
    assert rep.xheat_ave > 0
    Ts, Tk = out["spin_temperature"], out["kinetic_temp_neutral"]
    assert np.isfinite(Ts).all() and (Ts > 0).all() and (Tk > 0).all()
    xe = out["xray_ionised_fraction"]
    assert xe.min() >= 0 and xe.max() < 1
    # T_s lies between the two temperatures it is coupled to (T_c,eff sits next to T_k)
    lo = np.minimum(Tk, spec.Trad) * 0.9
    hi = np.maximum(Tk, spec.Trad) * 1.1
    assert np.mean((Ts >= lo) & (Ts <= hi)) > 0.99
    if not lagrangian:
        ave = np.array(rep.ave_sfrd[: spec.n_step])
        assert np.all(ave > 0)
        # avg_fix_term = mean_sfr_zpp / box mean of the table values = 1.1 (1 + 0.05 sin i)
        want = 1.1 * (1 + 0.05 * np.sin(np.arange(spec.n_step)))
        got = np.array(spec.mean_sfr_zpp[: spec.n_step]) / ave
        np.testing.assert_allclose(got, want, rtol=2e-3)  # np.interp vs the table's float knots


def test_heating_switches_change_the_temperature_only(oracle):
    spec, d = H.make(n=12, n_step=6)
    base = oracle.ts_grids(spec, d["density"], d["previous"], d["source"], d["filtered_density"])
    for flag in ("use_xray_heating", "use_cmb_heating", "use_lya_heating"):
        setattr(spec, flag, 0)
        off = oracle.ts_grids(spec, d["density"], d["previous"], d["source"], d["filtered_density"])
        setattr(spec, flag, 1)
        np.testing.assert_array_equal(off["xray_ionised_fraction"], base["xray_ionised_fraction"])
        diff = np.abs(off["kinetic_temp_neutral"] - base["kinetic_temp_neutral"])
        if flag == "use_cmb_heating":  # the recoil term as coded upstream (:1272-1279) is ~1e-9 K per unit z
            assert diff.max() < 1e-3
        else:
            assert diff.max() > 1e-3, flag


def test_init_first_Ts(oracle):
    fs = H.first_spec(n=12)
    rng = np.random.default_rng(1)
    dens = (0.05 * rng.standard_normal((12, 12, 12))).astype(np.float32)
    out = oracle.ts_first_grids(fs, dens)
    np.testing.assert_allclose(out["kinetic_temp_neutral"], fs.TK * (1 + fs.cT_ad * dens), rtol=3e-6)
    assert np.all(out["xray_ionised_fraction"] == np.float32(fs.xe))
    Trad = fs.T_cmb * (1 + fs.redshift)
    Ts = out["spin_temperature"]
    assert np.all((Ts > fs.TK) & (Ts < Trad))
    # denser cells couple more strongly to the gas
    i, j = np.unravel_index(dens.argmax(), dens.shape), np.unravel_index(dens.argmin(), dens.shape)
    assert Ts[i] < Ts[j]


def test_bad_requests_are_refused(oracle):
    spec, d = H.make(n=8, n_step=4)
    spec.lya_dEC = None
    with pytest.raises(RuntimeError):
        oracle.ts_grids(spec, d["density"], d["previous"], d["source"], None)
