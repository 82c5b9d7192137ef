"""CPU: the oracle's halo-catalogue branch of ComputeHaloBox (move_halo_galprops, map_mass.c:
346-476; set_halo_properties, HaloBox.c:62-102) against closed forms: one halo on a cell centre
puts property / cell volume into that cell; the relations without scatter are power laws with
exponential turnovers; totals are conserved by the CIC weights."""

import importlib

import numpy as np
import pytest

from halo_catalogue_helpers import attach, halo_consts, random_catalogue
from test_oracle_halobox import halobox_spec, make_tables, random_ics

S = importlib.import_module("21cmfast_amd.structs")
S_PER_YR = 31556925.9747


def still_ics(n, N, hires=False):
    ics = random_ics(n, N, hires, seed=1)
    for k in ics:
        if "_v" in k:
            ics[k][...] = 0
    return ics


def test_single_halo_closed_form(oracle):
    n = 8
    spec = halobox_spec(n, 2 * n, False, make_tables(), skip_integral=1)
    cell = spec.box_len / n
    M = 3.0e10
    cat = dict(masses=np.array([M], np.float32), coords=np.array([[2 * cell, 5 * cell, 7 * cell]], np.float32),
               star_rng=np.zeros(1, np.float32), sfr_rng=np.zeros(1, np.float32),
               xray_rng=np.zeros(1, np.float32))
    # no scatter, no upper turnover: f_* = f_*10 (M / 1e10)^a exp(-M_turn / M)
    c = halo_consts(sigma_star=0.0, sigma_sfr_lim=0.0, sigma_xray=0.0, upper_stellar_turnover=0)
    attach(spec, cat, c, skip_integral=True)
    out = oracle.halobox_grids(spec, still_ics(n, 2 * n), with_whalo=True, with_xray=True)
    fstar = c.fstar_10 * (M / 1e10) ** c.alpha_star * np.exp(-c.mturn_a_nofb / M)
    stars = fstar * M * c.baryon_ratio
    sfr = stars / (c.t_star * c.t_h)
    fesc = c.fesc_10 * (M / 1e10) ** c.alpha_esc
    vol_inv = 1.0 / cell**3
    want = {"n_ion": stars * c.pop2_ion * fesc * vol_inv, "halo_sfr": sfr * vol_inv,
            "whalo_sfr": sfr * c.pop2_ion * fesc * vol_inv,
            "halo_xray": c.l_x * sfr * S_PER_YR * vol_inv}  # L_X/SFR constant without the upper turnover
    for k, v in want.items():
        assert out[k][2, 5, 7] == pytest.approx(v, rel=3e-6), k
        assert np.count_nonzero(out[k]) == 1, k


def test_scatter_and_median(oracle):
    """A deviate of +1 sigma multiplies f_* by exp(sigma - sigma^2 / 2) (mean-preserving log-normal)
    and by exp(sigma) when the relations are medians (scaling_relations.c:350,381-383)."""
    n = 8
    M = 2.0e9
    cell = 1.5
    base = dict(masses=np.array([M], np.float32), coords=np.array([[cell, cell, cell]], np.float32),
                sfr_rng=np.zeros(1, np.float32), xray_rng=np.zeros(1, np.float32))
    vals = {}
    for key, rng, median in (("mean0", 0.0, 0), ("mean1", 1.0, 0), ("median0", 0.0, 1), ("median1", 1.0, 1)):
        spec = halobox_spec(n, 2 * n, False, make_tables())
        c = halo_consts(sigma_sfr_lim=0.0, scaling_median=median, use_xray=0)
        attach(spec, dict(base, star_rng=np.full(1, rng, np.float32)), c, skip_integral=True)
        vals[key] = float(oracle.halobox_grids(spec, still_ics(n, 2 * n))["halo_sfr"][1, 1, 1])
    sig = halo_consts().sigma_star
    assert vals["median1"] / vals["median0"] == pytest.approx(np.exp(sig), rel=1e-5)
    assert vals["mean1"] / vals["mean0"] == pytest.approx(np.exp(sig), rel=1e-5)
    assert vals["mean0"] / vals["median0"] == pytest.approx(np.exp(-sig * sig / 2), rel=1e-5)


@pytest.mark.parametrize("hires", [False, True])
def test_displacement_conserves_totals_and_adds_to_the_integral(oracle, hires):
    n, N = 10, 20
    tables = make_tables()
    cat = random_catalogue(3000, 1.5 * n, seed=5)
    ics = random_ics(n, N, hires, seed=4, vscale=6.0)
    halos_only = oracle.halobox_grids(attach(halobox_spec(n, N, hires, tables), cat, halo_consts(),
                                             skip_integral=True), ics, with_xray=True)
    frozen = oracle.halobox_grids(attach(halobox_spec(n, N, hires, tables), cat, halo_consts(),
                                         skip_integral=True), still_ics(n, N, hires), with_xray=True)
    for k in ("n_ion", "halo_sfr", "halo_xray"):
        assert halos_only[k].sum(dtype=np.float64) == pytest.approx(frozen[k].sum(dtype=np.float64), rel=2e-5)
        assert halos_only[k].min() >= 0 and not np.array_equal(halos_only[k], frozen[k])
    # with the integrated part: the sum of the two contributions (float adds: 1e-6)
    integral = oracle.halobox_grids(halobox_spec(n, N, hires, tables), ics)
    both = oracle.halobox_grids(attach(halobox_spec(n, N, hires, tables), cat, halo_consts()), ics)
    for k in ("n_ion", "halo_sfr"):
        np.testing.assert_allclose(both[k], halos_only[k].astype(np.float64) + integral[k], rtol=3e-5,
                                   atol=3e-6 * both[k].max())


def reference_kat_expectations(masses, rng, c, H_z):
    """The expectations of the reference's own known-answer test of the scaling relations
    (/root/reference/tests/test_halo_sampler.py:148-237, test_halo_prop_sampling: no upper
    turnover, no mini-halos, L_X/SFR constant), restated on the constants struct."""
    shmr = c.fstar_10 * (masses / 1e10) ** c.alpha_star * np.exp(
        -c.mturn_a_nofb / masses + rng * c.sigma_star - c.sigma_star**2 / 2)
    shmr = np.minimum(shmr, 1) * c.baryon_ratio
    sig = np.maximum(c.sigma_sfr_lim + c.sigma_sfr_idx * np.log10(shmr * masses / 1e10), c.sigma_sfr_lim)
    ssfr = H_z / c.t_star * np.exp(rng * sig - sig**2 / 2)
    lx = c.l_x * np.exp(rng * c.sigma_xray - c.sigma_xray**2 / 2)
    return shmr, ssfr, lx


def reference_kat_catalogue():
    masses, rng = np.meshgrid(np.array([1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12]),
                              np.array([-3.0, -2, -1, 0, 1, 2, 3]), indexing="ij")
    masses, rng = masses.ravel(), rng.ravel()
    return masses, rng, dict(masses=masses, coords=np.zeros((masses.size, 3)), star_rng=rng, sfr_rng=rng,
                             xray_rng=rng)


def test_scaling_relations_against_the_references_known_answers(oracle):
    """Pins set_halo_properties of the oracle to the reference's test_halo_prop_sampling: same halo
    masses (1e5..1e12), same deviates (-3..3 on the diagonal), same parameters (M_TURN 1e5,
    F_STAR10 0.1, ALPHA_STAR 0, t_STAR 0.1, L_X 1e40), same tolerances (1e-4; the reference allows
    3e-3 on the SSFR because its expectation takes H(z) from astropy)."""
    masses, rng, cat = reference_kat_catalogue()
    t_h = 1.0 / 5.3e-17
    c = halo_consts(10.0, upper_stellar_turnover=0, mturn_a_nofb=1e5, fstar_10=0.1, alpha_star=0.0,
                    t_star=0.1, t_h=t_h, l_x=1e40 * 1e-38)
    out = oracle.halo_props(c, cat, (8, 8, 8), 1.5, 10.0)
    shmr, ssfr, lx = reference_kat_expectations(masses, rng, c, 1.0 / t_h)
    np.testing.assert_array_equal(out[:, 0], masses.astype(np.float32))
    np.testing.assert_allclose(out[:, 1] / out[:, 0], shmr, rtol=1e-4)
    np.testing.assert_allclose(out[:, 2] / out[:, 1], ssfr, rtol=1e-4)
    np.testing.assert_allclose(out[:, 3] / (out[:, 2] * S_PER_YR), lx, rtol=1e-4)
    assert np.all(out[:, 8] == np.float32(1e5)) and np.all(out[:, 6:8] == 0) and np.all(out[:, 10] == 0)
