"""CPU: the oracle still reproduces the committed golden fixtures (tests/golden/*.npz, written by
tests/golden/make_golden.py).  Guards the checker itself against silent drift; the GPU twin is
tests/test_gpu_golden.py."""

import sys
from pathlib import Path

import numpy as np
import pytest

GOLDEN = Path(__file__).resolve().parent / "golden"
sys.path.insert(0, str(GOLDEN))
import cases  # noqa: E402


def check_ionize(out, gold, flag_tol, rtol, atol):
    ion_o, ion_g = out["neutral_fraction"] == 0, gold["neutral_fraction"] == 0
    assert np.mean(ion_o != ion_g) <= flag_tol
    same = ion_o == ion_g
    np.testing.assert_allclose(out["neutral_fraction"][same], gold["neutral_fraction"][same],
                               rtol=rtol, atol=atol)
    np.testing.assert_array_equal(out["z_reion"][same], gold["z_reion"][same])
    np.testing.assert_allclose(out["kinetic_temperature"][same], gold["kinetic_temperature"][same],
                               rtol=max(rtol, 1e-6), atol=2e4 * max(rtol, 1e-7))
    np.testing.assert_allclose(out["f_coll_grid_mean"], gold["f_coll_grid_mean"],
                               rtol=max(rtol / 10, 1e-10))
    assert float(out["global_xH"]) == pytest.approx(float(gold["global_xH"]), rel=max(rtol, 1e-9),
                                                    abs=2 * flag_tol)


@pytest.mark.parametrize("kind", ["lagrangian", "erfc"])
def test_oracle_ionize_matches_golden(oracle, kind):
    gold = np.load(GOLDEN / f"ionize_{kind}_{cases.N_ION}.npz")
    inp = {"density": gold["density"], "n_ion": gold["n_ion"]}
    out = cases.ionize_outputs(lambda s, d, n, nn: oracle.ionize_grids(s, d, n, need_nion=nn),
                               kind, inp)
    check_ionize(out, gold, flag_tol=1e-5, rtol=1e-6, atol=1e-8)
    assert 0.05 < (gold["neutral_fraction"] == 0).mean() < 0.95


def test_oracle_ionize_native_size_matches_golden(oracle):
    gold = np.load(GOLDEN / f"ionize_lagrangian_{cases.N_ION_NATIVE}.npz")
    inp = {"density": gold["density"], "n_ion": gold["n_ion"]}
    out = cases.ionize_outputs(lambda s, d, n, nn: oracle.ionize_grids(s, d, n, need_nion=nn),
                               "lagrangian", inp)
    check_ionize(out, gold, flag_tol=1e-5, rtol=1e-6, atol=1e-8)
    assert 0.05 < (gold["neutral_fraction"] == 0).mean() < 0.95


def test_oracle_perturb_roll_matches_golden(oracle):
    """The reference's one-cell-displacement test: the fixture holds the oracle's densities next
    to the analytic answer (rolled IC density x growth factor, atol 1e-3)."""
    gold = np.load(GOLDEN / "perturb_roll.npz")
    out = cases.perturb_roll_outputs(oracle.perturb_grids)
    for alg in (2, 1, 0):
        np.testing.assert_allclose(out[f"density_alg{alg}"], gold[f"density_alg{alg}"], atol=1e-6)
        np.testing.assert_allclose(gold[f"density_alg{alg}"], gold[f"expected_alg{alg}"], atol=1e-3)


def test_oracle_filters_match_golden(oracle):
    gold = np.load(GOLDEN / "filters_delta.npz")
    out = cases.filter_outputs(oracle.filter_grid)
    for k in gold.files:
        np.testing.assert_allclose(out[k], gold[k], rtol=1e-6, atol=1e-10, err_msg=k)
    # the reference's normalisation checks (tests/test_filtering.py:160-236)
    for ft in (0, 1, 2, 4):
        assert float(gold[f"filter{ft}_sum"]) == pytest.approx(1.0, abs=1e-4)
    q = cases.FILTER_PARAM[3] / cases.FILTER_RADII[3]
    want = 6 * q**3 - np.exp(-1 / q) * (6 * q**3 + 6 * q**2 + 3 * q)
    assert float(gold["filter3_sum"]) == pytest.approx(want, abs=1e-4)


def test_oracle_ics_perturb_match_golden(oracle):
    gold = np.load(GOLDEN / "ics_perturb.npz")
    out = cases.ics_perturb_outputs(oracle.new_ics_arrays, oracle.ics_grids, oracle.perturb_grids,
                                    gold["hires_density"])
    for k, v in out.items():
        scale = np.abs(gold[k]).max()
        np.testing.assert_allclose(v, gold[k], rtol=1e-6, atol=1e-6 * scale, err_msg=k)


def test_oracle_tsfilter_matches_golden(oracle):
    gold = np.load(GOLDEN / "tsfilter.npz")
    inp = {k: gold[k] for k in ("ts_density", "ts_sfr", "ts_xray")}
    out = cases.tsfilter_outputs(oracle.fill_Rbox_grids, oracle.annular_filter_grids, inp)
    for k, v in out.items():
        scale = np.abs(gold[k]).max()
        np.testing.assert_allclose(v, gold[k], rtol=1e-6, atol=1e-6 * scale, err_msg=k)
    # the multiple-scattering shell differs from the straight-line one, the x-ray grid (window 4
    # in both) does not
    assert not np.allclose(gold["shell_sl_sfr"], gold["shell_ms_sfr"], atol=1e-4)
    np.testing.assert_array_equal(gold["shell_sl_xray"], gold["shell_ms_xray"])
