"""CPU: the oracle's recombination branch (reference: IonisationBox.c:583-663,1084-1140,
1258-1340, recombinations.c:64-92) through properties that follow from the reference's text."""

import numpy as np
import pytest

from recomb_helpers import DZ, inputs, recomb_spec, synthetic_rr_tables


def rr_numpy(z_eff, gamma):
    """splined_recombination_rate via scipy's natural spline on the same table."""
    from scipy.interpolate import CubicSpline
    from recomb_helpers import LNG_MIN, S, ln_gamma_knots

    y, _ = synthetic_rr_tables()
    lnG = ln_gamma_knots()
    z_ct = min(max(int(z_eff / DZ + 0.5), 0), S.RR_NZ - 1)
    if gamma <= 0 or np.log(gamma) < LNG_MIN:
        return 0.0
    return float(CubicSpline(lnG, y[z_ct], bc_type="natural")(min(np.log(gamma), lnG[-1] - 1e-7)))


def test_spline_evaluation_matches_scipy(oracle):
    import ctypes as C

    lib = oracle.load()
    lib.oracle_splined_recombination_rate.restype = C.c_double
    lib.oracle_splined_recombination_rate.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_double]
    y, c = synthetic_rr_tables()
    for z, g in ((8.3, 0.2), (0.05, 1e-3), (59.9, 30.0), (12.0, 5e-5), (7.0, 1e9), (6.1, 1.0)):
        got = lib.oracle_splined_recombination_rate(y.ctypes.data, c.ctypes.data, z, g)
        assert got == pytest.approx(rr_numpy(z, g), rel=1e-10, abs=1e-14)
    assert lib.oracle_splined_recombination_rate(y.ctypes.data, c.ctypes.data, 8.0, 1e-5) == 0.0


@pytest.mark.parametrize("lagrangian", [True, False])
def test_first_crossing_outputs(oracle, lagrangian):
    n = 24
    spec = recomb_spec(n, model=2, cell_recomb=1, lagrangian=lagrangian)
    d = inputs((n, n, n))
    zero = np.zeros_like(d["prev_nrec"])
    kw = dict(n_ion=d["n_ion"], whalo_sfr=d["whalo_sfr"]) if lagrangian else dict(need_nion=True)
    out = oracle.ionize_grids(spec, d["density"], prev_nrec=zero,
                              prev_z_reion=d["prev_z_reion"], **kw)
    # N_rec = 0 everywhere: the barrier is the recombination-free one
    base = recomb_spec(n, lagrangian=lagrangian)
    base.recomb_model = 0
    ref = oracle.ionize_grids(base, d["density"], prev_z_reion=d["prev_z_reion"],
                              **{k: v for k, v in kw.items() if k != "whalo_sfr"})
    np.testing.assert_array_equal(out["neutral_fraction"], ref["neutral_fraction"])
    ion = out["neutral_fraction"] == 0
    assert 0.05 < ion.mean() < 0.95
    g12, mfp, nrec = (out[k] for k in ("ionisation_rate_G12", "mean_free_path",
                                       "cumulative_recombinations"))
    # Gamma_12 and the mean free path are written exactly where a barrier was crossed
    assert np.all(g12[~ion] == 0) and np.all(mfp[~ion] == 0)
    assert np.all(g12[ion] >= 0) and (g12[ion] > 0).mean() > 0.9
    radii = np.array([spec.R[i] for i in range(spec.n_radii)], np.float32)
    assert np.isin(mfp[ion], radii).all()
    # ... and the mean free path is the LARGEST radius that ionises the cell: rerunning with the
    # ladder cut below radius index k leaves cells with mfp >= R_k ionised
    k = spec.n_radii // 2
    cut = recomb_spec(n, lagrangian=lagrangian)
    cut.r_lowest = k
    part = oracle.ionize_grids(cut, d["density"], prev_nrec=zero, prev_z_reion=d["prev_z_reion"],
                               **kw)
    np.testing.assert_array_equal(part["neutral_fraction"] == 0, mfp >= radii[k])
    # set_recombination_rates: N_rec = prev + rate(z_eff - 1, Gamma_12) |dt/dz| dz (1 - x_HI)
    i = tuple(np.argwhere(ion & (g12 > np.exp(-9)))[0])
    z_eff = (1.0 + d["density"][i]) ** (1.0 / 3.0) * (1 + spec.stored_redshift)
    want = rr_numpy(z_eff - 1.0, float(g12[i])) * spec.fabs_dtdz * spec.dz * 1.0
    assert nrec[i] == pytest.approx(want, rel=1e-5)
    assert np.all(nrec[~ion & (out["neutral_fraction"] == 1)] == 0)


def test_recombinations_delay_reionisation(oracle):
    n = 24
    d = inputs((n, n, n))
    frac = []
    for scale in (0.0, 1.0, 6.0):
        for cell in (1, 0):
            spec = recomb_spec(n, model=2, cell_recomb=cell)
            out = oracle.ionize_grids(spec, d["density"], d["n_ion"], whalo_sfr=d["whalo_sfr"],
                                      prev_nrec=d["prev_nrec"] * np.float32(scale),
                                      prev_z_reion=d["prev_z_reion"])
            frac.append((out["neutral_fraction"] == 0).mean())
            # the accumulated count never decreases
            assert np.all(out["cumulative_recombinations"] >=
                          d["prev_nrec"] * np.float32(scale) - 1e-7)
    assert frac[0] == frac[1]          # no recombinations: filtering zeros changes nothing
    assert frac[2] < frac[0] and frac[4] < frac[2]
    assert frac[3] < frac[1] and frac[5] < frac[3]


def test_homogeneous_model(oracle):
    n = 20
    spec = recomb_spec(n, model=1, cell_recomb=1)
    d = inputs((n, n, n))
    prev = np.full((1, 1, 1), 0.3, np.float32)
    out = oracle.ionize_grids(spec, d["density"], d["n_ion"], whalo_sfr=d["whalo_sfr"],
                              prev_nrec=prev, prev_z_reion=d["prev_z_reion"])
    assert out["cumulative_recombinations"].shape == (1, 1, 1)
    g = float(np.float32(out["ionisation_rate_G12"].astype(np.float64).mean()))
    xh = float(np.float32(out["report"].global_xH))
    want = 0.3 + rr_numpy(spec.stored_redshift, g) * spec.fabs_dtdz * spec.dz * (1 - xh)
    assert float(out["cumulative_recombinations"][0, 0, 0]) == pytest.approx(want, rel=1e-6)
    # a uniform N_rec = 0.3 in every cell (inhomogeneous, CELL_RECOMB) gives the same barrier
    spec2 = recomb_spec(n, model=2, cell_recomb=1)
    out2 = oracle.ionize_grids(spec2, d["density"], d["n_ion"], whalo_sfr=d["whalo_sfr"],
                               prev_nrec=np.full((n, n, n), 0.3, np.float32),
                               prev_z_reion=d["prev_z_reion"])
    np.testing.assert_array_equal(out["neutral_fraction"], out2["neutral_fraction"])
    np.testing.assert_array_equal(out["ionisation_rate_G12"], out2["ionisation_rate_G12"])
