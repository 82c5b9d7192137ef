"""CPU: the oracle's ComputeHaloBox with USE_MINI_HALOS -- cell values from the 2-D tables with
the cell's turnover masses (HaloBox.c:245-283, map_mass.c:285-321) and get_log10_turnovers with
upstream's per-thread running maximum of the atomic turnover (HaloBox.c:465-516)."""
import importlib

import numpy as np
import pytest

import halobox_mini_helpers as HM
from test_oracle_halobox import random_ics

S = importlib.import_module("21cmfast_amd.structs")


def lerp2(tab, x, y, x0, xw, y0, yw):
    i = np.floor((x - x0) / xw).astype(int)
    j = np.floor((y - y0) / yw).astype(int)
    px, py = (x - (x0 + xw * i)) / xw, (y - (y0 + yw * j)) / yw
    t = tab.astype(np.float64)
    return ((t[i, j] * (1 - py) + t[i, j + 1] * py) * (1 - px)
            + (t[i + 1, j] * (1 - py) + t[i + 1, j + 1] * py) * px)


def test_no_displacement_cell_values(oracle):
    n = 12
    spec = HM.mini_spec(n)
    ics = random_ics(n, n, False, seed=2)
    for k in list(ics):
        if "_v" in k:
            ics[k][...] = 0
    got = oracle.halobox_grids(spec, ics, with_whalo=True, with_xray=True)
    x = ics["lowres_density"].astype(np.float64) * spec.growth_factor
    mta, mtm = (a.astype(np.float64) for a in spec._mt)
    k = spec._keep
    tab_na, tab_nm, tab_sm, tab_x = k[-4], k[-3], k[-2], k[-1]
    na = np.exp(lerp2(tab_na, x, mta, spec.tab_min, spec.tab_width, spec.mta_min, spec.mta_width))
    nm = np.exp(lerp2(tab_nm, x, mtm, spec.tab_min, spec.tab_width, spec.mtm_min, spec.mtm_width))
    sm = np.exp(lerp2(tab_sm, x, mtm, spec.tab_min, spec.tab_width, spec.mt_fixed_min, spec.mt_fixed_width))
    lx = np.exp(lerp2(tab_x, x, mtm, spec.tab_min, spec.tab_width, spec.mt_fixed_min, spec.mt_fixed_width))
    np.testing.assert_allclose(got["n_ion"], na * spec.prefactor_nion + nm * spec.prefactor_nion_mini,
                               rtol=3e-6)
    np.testing.assert_allclose(got["halo_sfr_mini"], sm * spec.prefactor_sfr_mini, rtol=3e-6)
    np.testing.assert_allclose(got["halo_xray"], lx * spec.prefactor_xray, rtol=3e-6)
    np.testing.assert_allclose(got["whalo_sfr"], got["n_ion"] * np.float32(spec.prefactor_wsfr), rtol=3e-7)
    # a displaced run conserves the totals of every grid
    ics2 = random_ics(n, n, False, seed=2)
    moved = oracle.halobox_grids(spec, ics2, with_whalo=True, with_xray=True)
    for f in ("n_ion", "halo_sfr", "halo_sfr_mini", "halo_xray"):
        assert moved[f].astype(np.float64).sum() == pytest.approx(got[f].astype(np.float64).sum(), rel=2e-5)


def test_turnovers_running_maximum(oracle):
    shape = (10, 10, 12)
    spec, g12, zre, j21, vcb = HM.turnover_inputs(shape)
    m_turn = 10 ** 5.0
    z = np.float32(11.0)
    with np.errstate(divide="ignore", invalid="ignore"):
        m_re = np.where(zre <= 1e-19, 1e-40,
                        3e9 * (2.0 * g12.astype(np.float64)) ** 0.17 * ((1.0 + z) / 10) ** -2.1
                        * np.maximum(1 - ((1.0 + z) / (1.0 + zre.astype(np.float64))) ** 2, 0) ** 2.5)
    m_lw = (3.314e7 * (1.0 + z) ** -1.5 * (1 + 2.0 * j21.astype(np.float64) ** 0.6)
            * (1 + vcb.astype(np.float64) / spec.sigma_vcb) ** 1.8)
    per_cell = np.maximum(m_re, m_turn)
    for T in (1, 4, 7):
        a, m, ave = oracle.halobox_turnovers(spec, m_turn, 1, T, g12, zre, j21, vcb)
        np.testing.assert_allclose(m, np.log10(np.maximum(m_lw, per_cell)), rtol=2e-7)
        # atomic turnover: running maximum inside each of the T contiguous shares
        flat = per_cell.ravel()
        N = flat.size
        q, r = divmod(N, T)
        want = np.empty(N)
        s = 0
        for t in range(T):
            ln = q + (1 if t < r else 0)
            want[s:s + ln] = np.maximum.accumulate(np.maximum(flat[s:s + ln], spec.mturn_a_nofb))
            s += ln
        np.testing.assert_allclose(a.ravel(), np.log10(want), rtol=2e-7)
        assert ave[0] == pytest.approx(a.astype(np.float64).mean(), rel=1e-6)
        assert ave[1] == pytest.approx(m.astype(np.float64).mean(), rel=1e-6)
    a1 = oracle.halobox_turnovers(spec, m_turn, 1, 1, g12, zre, j21, vcb)[0]
    a7 = oracle.halobox_turnovers(spec, m_turn, 1, 7, g12, zre, j21, vcb)[0]
    assert (a1 >= a7).all() and (a1 > a7).any()  # fewer threads: the maximum carries further
    # above Z_HEAT_MAX nothing has formed: no feedback, no LW background (:488-492)
    a0, m0, _ = oracle.halobox_turnovers(spec, m_turn, 0, 4, None, None, None, vcb, shape=shape)
    assert (a0 == np.float32(np.log10(spec.mturn_a_nofb))).all()
    np.testing.assert_allclose(
        m0, np.log10(np.maximum(3.314e7 * 12.0 ** -1.5 * (1 + vcb.astype(np.float64) / spec.sigma_vcb) ** 1.8,
                                m_turn)), rtol=2e-7)
