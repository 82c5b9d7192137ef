"""CPU: the oracle's ComputeHaloBox integrated branch (move_grid_galprops, map_mass.c:214-344)
against closed-form cases: without displacement every Lagrangian cell deposits
exp(lerp(ln-table, delta D)) * prefactor into itself; a uniform displacement of whole output
cells rolls the grids; any displacement conserves the totals."""

import importlib

import numpy as np
import pytest

S = importlib.import_module("21cmfast_amd.structs")

D_Z, D_ZI = 0.12, 0.0042


def make_tables(tab_min=-0.9, tab_max=1.6):
    x = np.linspace(tab_min, tab_max, S.NDELTA_TABLE)
    ln_nion = (-9.0 + 4.5 * x - 0.7 * x * x).astype(np.float32)
    ln_sfrd = (-6.0 + 3.8 * x - 0.5 * x * x).astype(np.float32)
    return tab_min, (tab_max - tab_min) / (S.NDELTA_TABLE - 1.0), ln_nion, ln_sfrd


def xray_table(tables):
    """A third ln-table (X-ray emissivity) on the grid of make_tables()."""
    tab_min, tab_width = tables[0], tables[1]
    x = tab_min + tab_width * np.arange(S.NDELTA_TABLE)
    return (-2.0 + 4.1 * x - 0.6 * x * x).astype(np.float32)


def with_xray(spec, tables, prefactor=4.4e3):
    ln_xray = xray_table(tables)
    spec.ln_xray_table = ln_xray.ctypes.data_as(S.c_float_p)
    spec.prefactor_xray = prefactor
    spec._keep = tuple(spec._keep) + (ln_xray,)
    return spec


def halobox_spec(n, N, hires, tables, lpt2=1, **kw):
    tab_min, tab_width, ln_nion, ln_sfrd = tables
    spec = S.HaloBoxSpec(dim=N, dim_z=N, hii_dim=n, hii_dim_z=n, box_len=1.5 * n, box_len_z=1.5 * n,
                         perturb_on_high_res=int(hires), lpt2=lpt2, growth_factor=D_Z,
                         init_growth_factor=D_ZI, tab_min=tab_min, tab_width=tab_width,
                         ln_nion_table=ln_nion.ctypes.data_as(S.c_float_p),
                         ln_sfrd_table=ln_sfrd.ctypes.data_as(S.c_float_p),
                         prefactor_nion=3.1e9, prefactor_sfr=2.2e-7, prefactor_wsfr=0.37)
    spec._keep = (ln_nion, ln_sfrd)
    return spec.update(**kw) if kw else spec


def random_ics(n, N, hires, seed, vscale=1.0):
    rng = np.random.default_rng(seed)
    pre, m = ("hires", N) if hires else ("lowres", n)
    ics = {f"{pre}_density": (3.0 * rng.standard_normal((m,) * 3)).astype(np.float32).clip(-7, 12)}
    for ax in "xyz":
        ics[f"{pre}_v{ax}"] = (vscale * 1.5 * rng.standard_normal((m,) * 3)).astype(np.float32)
        ics[f"{pre}_v{ax}_2LPT"] = (vscale * 0.8 * rng.standard_normal((m,) * 3)).astype(np.float32)
    return ics


def cell_values(spec, dens, tables):
    tab_min, tab_width, ln_nion, ln_sfrd = tables
    x = dens.astype(np.float64) * spec.growth_factor
    idx = np.floor((x - tab_min) / tab_width).astype(int)
    t = (x - (tab_min + tab_width * idx.astype(np.float32).astype(np.float64))) / tab_width
    lerp = lambda y: y[idx].astype(np.float64) * (1 - t) + y[idx + 1].astype(np.float64) * t  # noqa: E731
    return np.exp(lerp(ln_nion)) * spec.prefactor_nion, np.exp(lerp(ln_sfrd)) * spec.prefactor_sfr


def test_no_displacement_and_roll(oracle):
    n = 12
    tables = make_tables()
    spec = halobox_spec(n, 2 * n, False, tables)
    ics = random_ics(n, 2 * n, False, seed=1)
    for k in list(ics):
        if "_v" in k:
            ics[k][...] = 0
    nion, sfr = cell_values(spec, ics["lowres_density"], tables)
    out = oracle.halobox_grids(spec, ics, with_whalo=True)
    np.testing.assert_allclose(out["n_ion"], nion, rtol=2e-7)
    np.testing.assert_allclose(out["halo_sfr"], sfr, rtol=2e-7)
    np.testing.assert_allclose(out["whalo_sfr"], out["n_ion"].astype(np.float64) * 0.37, rtol=2e-7)
    # one cell along +y at first order, one cell along -z from the 2LPT term (test_perturb.py:52-106)
    cell = spec.box_len / n
    ics["lowres_vy"][...] = cell / (D_Z - D_ZI)
    ics["lowres_vz_2LPT"][...] = cell / ((-3.0 / 7.0) * (D_Z**2 - D_ZI**2))
    out = oracle.halobox_grids(spec, ics)
    np.testing.assert_allclose(out["n_ion"], np.roll(nion, (0, 1, -1), (0, 1, 2)), rtol=3e-5,
                               atol=2e-6 * nion.max())  # float velocities: ~1e-7 of a cell leaks


@pytest.mark.parametrize("hires", [False, True])
def test_totals_are_conserved(oracle, hires):
    n, N = 10, 20
    tables = make_tables()
    spec = halobox_spec(n, N, hires, tables)
    ics = random_ics(n, N, hires, seed=4, vscale=6.0)
    dens = ics["hires_density" if hires else "lowres_density"]
    nion, sfr = cell_values(spec, dens, tables)
    out = oracle.halobox_grids(spec, ics)
    assert out["n_ion"].astype(np.float64).sum() == pytest.approx(nion.sum(), rel=2e-5)
    assert out["halo_sfr"].astype(np.float64).sum() == pytest.approx(sfr.sum(), rel=2e-5)
    assert out["n_ion"].min() >= 0


def test_xray_grid_follows_its_own_table(oracle):
    """USE_TS_FLUCT: the third value per Lagrangian cell, exp(lerp(ln-xray table)) * prefactor,
    deposited like the other two (HaloBox.c:279-283, map_mass.c:316-319)."""
    n = 12
    tables = make_tables()
    spec = with_xray(halobox_spec(n, 2 * n, False, tables), tables)
    ics = random_ics(n, 2 * n, False, seed=4)
    for k in list(ics):
        if "_v" in k:
            ics[k][...] = 0
    out = oracle.halobox_grids(spec, ics, with_xray=True)
    x = ics["lowres_density"].astype(np.float64) * spec.growth_factor
    tab_min, tab_width = tables[0], tables[1]
    idx = np.floor((x - tab_min) / tab_width).astype(int)
    t = (x - (tab_min + tab_width * idx.astype(np.float32).astype(np.float64))) / tab_width
    y = xray_table(tables)
    want = np.exp(y[idx].astype(np.float64) * (1 - t) + y[idx + 1].astype(np.float64) * t) * 4.4e3
    np.testing.assert_allclose(out["halo_xray"], want, rtol=2e-6)
    # the other grids do not change when the X-ray grid is requested
    base = oracle.halobox_grids(halobox_spec(n, 2 * n, False, tables), ics)
    np.testing.assert_array_equal(out["n_ion"], base["n_ion"])
    np.testing.assert_array_equal(out["halo_sfr"], base["halo_sfr"])
    # displaced: totals conserved
    ics2 = random_ics(n, 2 * n, False, seed=4)
    moved = oracle.halobox_grids(spec, ics2, with_xray=True)
    assert moved["halo_xray"].sum(dtype=np.float64) == pytest.approx(want.sum(), rel=1e-5)
