"""The oracle's spin-temperature cell algorithm with the molecularly cooled population
(USE_MINI_HALOS, E-INTEGRAL): consistency with the one-population path (pinned to the reference's
fixtures) and the relations the algorithm guarantees.
Reference behaviour: SpinTemperatureBox.c:535-565,1011-1075,1642-1733,1843-1845,1324."""
import importlib
import math

import numpy as np
import pytest

import ts_helpers as T

S = importlib.import_module("21cmfast_amd.structs")


def run(oracle, spec, inp):
    return oracle.ts_grids(spec, inp["density"], inp["previous"], inp["source"],
                           inp["filtered_density"])


def test_dark_mini_population_changes_nothing(oracle):
    spec0, inp0 = T.make(n=16, n_step=8, lagrangian=False)
    ref = run(oracle, spec0, inp0)
    spec, inp = T.make(n=16, n_step=8, lagrangian=False)
    T.add_minis(spec, inp, strength=0.0)
    for i in range(spec.n_step):  # no Pop-II Lyman-Werner term either: J_21_LW must vanish
        spec.lw_prefactor[i] = 0.0
    got = run(oracle, spec, inp)
    for k in ("spin_temperature", "kinetic_temp_neutral", "xray_ionised_fraction"):
        np.testing.assert_array_equal(got[k], ref[k])
    assert (got["J_21_LW"] == 0).all()


def test_lyman_werner_background_and_heating(oracle):
    spec, inp = T.make(n=16, n_step=8, lagrangian=False)
    base = run(oracle, spec, inp)
    T.add_minis(spec, inp)
    got = run(oracle, spec, inp)
    rep = got["report"]
    n_step = spec.n_step
    fd, mc = inp["filtered_density"], inp["filtered_log10_mcrit"]
    # J_21_LW from the analytic tables behind add_minis / make (cells away from the -50 floor)
    lw = np.zeros(fd.shape[1:])
    for i in range(n_step):
        g = spec.zpp_growth[i]
        d = fd[i].astype(np.float64) * g
        fa = np.exp(np.maximum(-9.0 + 4.0 * d - 0.5 * d * d - 0.1 * i, -50.0))
        fm = np.exp(-10.0 + 3.0 * d - 0.4 * d * d - 0.1 * i - 1.2 * (mc[i].astype(np.float64) - 6.0))
        assert rep.ave_sfrd_mini[i] == pytest.approx(fm.mean(), rel=2e-3)
        sa = ((1 + d) * fa).astype(np.float32) * spec.z_edge_factor[i] * (
            spec.mean_sfr_zpp[i] / rep.ave_sfrd[i]) * spec.sfr_scale
        sm = ((1 + d) * fm).astype(np.float32) * spec.z_edge_factor[i] * (
            spec.mean_sfr_zpp_mini[i] / rep.ave_sfrd_mini[i]) * spec.sfr_scale_mini
        lw += sa * spec.lw_prefactor[i] + sm * spec.lw_prefactor_mini[i]
    want = lw * spec.lya_star_prefactor * spec.volunit_inv * spec.h_p * 1e21
    np.testing.assert_allclose(got["J_21_LW"], want, rtol=5e-3)
    assert 1e-4 < got["J_21_LW"].mean() < 1e4
    # more sources: more X-ray heating / ionisation and a stronger Lyman-alpha coupling
    assert got["report"].xion_ave > base["report"].xion_ave
    assert got["report"].J_alpha_ave > base["report"].J_alpha_ave
    assert got["xray_ionised_fraction"].mean() > base["xray_ionised_fraction"].mean()


def test_source_grids_with_mini_halos(oracle):
    """Lagrangian source grids: J_21_LW from the analytic sums, the straight-line copies replace
    the scattering-filtered grids in the Lyman-Werner term only."""
    spec, inp = T.make(n=14, n_step=7, lagrangian=True)
    base = run(oracle, spec, inp)
    T.add_minis_grids(spec, inp)
    got = run(oracle, spec, inp)
    src = inp["source"]
    scale = spec.lya_star_prefactor * spec.volunit_inv * spec.h_p * 1e21
    lw = sum((src["filtered_sfr"][i].astype(np.float64) * spec.lw_prefactor[i]
              + src["filtered_sfr_mini"][i].astype(np.float64) * spec.lw_prefactor_mini[i])
             * spec.z_edge_factor[i] for i in range(spec.n_step))
    np.testing.assert_allclose(got["J_21_LW"], lw * scale, rtol=3e-6)
    # X-rays come from filtered_xray alone (both populations are in it upstream): x_e unchanged;
    # the extra Lyman-alpha photons couple T_s closer to T_k
    np.testing.assert_array_equal(got["xray_ionised_fraction"], base["xray_ionised_fraction"])
    assert got["report"].J_alpha_ave > base["report"].J_alpha_ave
    spec2, inp2 = T.make(n=14, n_step=7, lagrangian=True)
    T.add_minis_grids(spec2, inp2, lw_copies=True)
    got2 = run(oracle, spec2, inp2)
    s2 = inp2["source"]
    lw2 = sum((s2["filtered_sfr_lw"][i].astype(np.float64) * spec2.lw_prefactor[i]
               + s2["filtered_sfr_mini_lw"][i].astype(np.float64) * spec2.lw_prefactor_mini[i])
              * spec2.z_edge_factor[i] for i in range(spec2.n_step))
    np.testing.assert_allclose(got2["J_21_LW"], lw2 * scale, rtol=3e-6)
    np.testing.assert_array_equal(got2["spin_temperature"], got["spin_temperature"])


def test_mcrit_grid(oracle):
    shape = (10, 10, 14)
    rng = np.random.default_rng(3)
    ms = S.MturnSpec(hii_dim=10, hii_dim_z=14, redshift=15.0, vcb_const=20.0, A_LW=2.0, BETA_LW=0.6,
                     A_VCB=1.0, BETA_VCB=1.8, sigma_vcb=25.86 * math.sqrt(3 * math.pi / 8))
    j21 = (2.0 * rng.random(shape) ** 3).astype(np.float32)
    vcb = (40 * rng.random(shape)).astype(np.float32)
    for v in (vcb, None):
        got = oracle.ts_mcrit_grid(ms, 10 ** 5.0, j21, v)
        vv = (vcb if v is not None else np.float32(20.0)).astype(np.float64)
        want = np.log10(np.maximum(3.314e7 * 16.0 ** -1.5 * (1 + 2.0 * j21.astype(np.float64) ** 0.6)
                                   * (1 + vv / ms.sigma_vcb) ** 1.8, 1e5))
        np.testing.assert_allclose(got, want, rtol=2e-7)
    # a high atomic turnover floors the grid
    assert (oracle.ts_mcrit_grid(ms, 10 ** 9.0, j21, vcb) == np.float32(9.0)).all()


def test_no_light_and_refusals(oracle):
    spec, inp = T.make(n=12, n_step=6, lagrangian=False, no_light=True)
    T.add_minis(spec, inp)
    got = run(oracle, spec, inp)
    assert (got["J_21_LW"] == 0).all() and np.isfinite(got["spin_temperature"]).all()
    spec, inp = T.make(n=12, n_step=6, lagrangian=True)
    spec.use_mini_halos = 1  # the mini tables belong to the Eulerian E-INTEGRAL mode
    with pytest.raises(RuntimeError):
        run(oracle, spec, inp)
