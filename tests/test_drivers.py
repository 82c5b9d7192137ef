"""Driver-layer bookkeeping of the Lagrangian spin-temperature path (21cmfast_amd/drivers.py;
reference: src/py21cmfast/drivers/single_field.py:382-636).  No GPU needed."""

import importlib
import math

import numpy as np
import pytest
from scipy import integrate

D = importlib.import_module("21cmfast_amd.drivers")


def test_flat_cosmology_is_astropy_planck18():
    """astropy.cosmology.Planck18 (H0 = 67.66, Om0 = 0.30966, Tcmb0 = 2.7255, Neff = 3.046,
    m_nu = [0, 0, 0.06] eV): published derived values of that realisation."""
    c = D.FlatCosmology(0.6766, 0.30966)
    assert c.Ogamma0 == pytest.approx(5.4020e-5, rel=2e-4)
    assert c.Onu0 == pytest.approx(1.4397e-3, rel=5e-4)
    assert c.Ode0 == pytest.approx(0.68885, rel=2e-5)
    assert c.efunc(0.0) == pytest.approx(1.0, rel=1e-12)
    # the comoving distance against an independent adaptive quadrature of c / H(z)
    for z in (0.5, 6.0, 18.0, 35.0):
        want = D.C_KMS / 67.66 * integrate.quad(lambda x: 1 / c.efunc(x), 0, z, epsrel=1e-11)[0]
        assert c.comoving_distance(z) == pytest.approx(want, rel=1e-9)
    assert c.comoving_distance(1.0) == pytest.approx(3395.6, rel=1e-4)  # Planck18.comoving_distance(1)
    d = c.comoving_distance(np.array([10.0, 20.0]))
    assert d.shape == (2,) and d[1] > d[0]
    assert c.z_at_comoving_distance(c.comoving_distance(17.3)) == pytest.approx(17.3, rel=1e-9)


def test_shells_follow_the_light_cone():
    c = D.FlatCosmology(0.6766, 0.30966)
    R, zavg = D.xray_shells(18.0, 50, 100.0, 40, 500.0, c)
    assert R[0] == pytest.approx(D.L_FACTOR * 2.0) and R[-1] < 500.0
    np.testing.assert_allclose(R[1:] / R[:-1], (500.0 / R[0]) ** (1 / 40), rtol=1e-12)
    assert np.all(np.diff(zavg) > 0) and zavg[0] > 18.0
    # shell i ends R[i] comoving Mpc behind z = 18; its mean redshift is the midpoint in z
    edges = np.array([c.z_at_comoving_distance(c.comoving_distance(18.0) + r) for r in R])
    want = edges - np.diff(np.insert(edges, 0, 18.0)) / 2
    np.testing.assert_allclose(zavg, want, rtol=2e-6)  # the reference interpolates on 100 z points
    # HII_DIM = 1 (global evolution): 1.5 Mpc cells
    R1, _ = D.xray_shells(18.0, 1, 1e6, 40, 500.0, c)
    assert R1[0] == pytest.approx(1.5 * D.L_FACTOR)


def test_halo_box_interpolation():
    z = [18.0, 18.76, 19.55]
    boxes = [{"halo_sfr": np.full((2, 2, 2), v, np.float32), "halo_xray": np.full((2, 2, 2), 10 * v, np.float32)}
             for v in (3.0, 2.0, 1.0)]
    out = D.interp_halo_boxes(z, boxes, ("halo_sfr", "halo_xray"), 18.38)
    t = (18.38 - 18.0) / 0.76
    np.testing.assert_allclose(out["halo_sfr"], (1 - t) * 3.0 + t * 2.0, rtol=1e-6)
    np.testing.assert_allclose(out["halo_xray"], 10 * ((1 - t) * 3.0 + t * 2.0), rtol=1e-6)
    assert out["halo_sfr"].dtype == np.float32
    # on a node: searchsorted(side="left") takes it as the upper end of the lower interval
    np.testing.assert_allclose(D.interp_halo_boxes(z, boxes, ("halo_sfr",), 18.76)["halo_sfr"], 2.0)
    for bad in (17.9, 19.6, 18.0):  # outside, or on the lowest node (idx_prog == 0 upstream)
        with pytest.raises(ValueError):
            D.interp_halo_boxes(z, boxes, ("halo_sfr",), bad)
    with pytest.raises(ValueError, match="ascending"):
        D.interp_halo_boxes(z[::-1], boxes, ("halo_sfr",), 18.3)


def test_lya_diffusion_scale():
    """Eq. 24 of arXiv:2601.14360 as coded upstream: R_star grows with (1 + z) x_HI; ~ 20 comoving
    Mpc at z = 18 in a neutral universe (the value the multiple-scattering fixture pin runs with)."""
    c = D.FlatCosmology(0.6766, 0.30966)
    r = D.lya_diffusion_scale(18.0, 1.0, 0.6766, 0.30966, 0.04897, 0.24, c)
    assert 5.0 < r < 80.0
    assert D.lya_diffusion_scale(18.0, 0.5, 0.6766, 0.30966, 0.04897, 0.24, c) == pytest.approx(0.5 * r)
    assert D.lya_diffusion_scale(37.0, 1.0, 0.6766, 0.30966, 0.04897, 0.24, c) == pytest.approx(2 * r)
    # the closed form
    n_H = 0.76 * c.rho_crit0 * 0.04897 / D.M_P
    want = 3 * D.C_CMS**4 * 6.25e8**2 * n_H * 19.0 / (32 * math.pi**3 * 2.46606727e15**4 * c.H0_cgs**2 * 0.30966)
    assert r == pytest.approx(want / D.MPC_CM, rel=1e-12)


def test_required_redshifts_insert_requested_snapshots_between_nodes():
    """a requested redshift between two nodes is computed AT that redshift and never
    becomes a "previous" box; two requests near one node do not overwrite each other
    (reference: drivers/coeval.py:971-992, 880-884)."""
    i = D.Inputs(SOURCE_MODEL=1, USE_TS_FLUCT=True, Z_HEAT_MAX=20.0, ZPRIME_STEP_FACTOR=1.1)
    allz, nodes = D.required_redshifts(i, [12.0, 12.3, 9.0])
    assert allz == sorted(allz, reverse=True) and len(set(allz)) == len(allz)
    for z in (12.0, 12.3, 9.0):
        assert float(np.float32(z)) in allz
    assert float(np.float32(12.3)) not in nodes and float(np.float32(12.0)) not in nodes
    # the nodes are the log-spaced ladder from the lowest request up to Z_HEAT_MAX
    ladder = sorted(nodes, reverse=True)
    np.testing.assert_allclose(np.diff(np.log1p(ladder)), -np.log(1.1), rtol=1e-6)
    assert ladder[0] >= 20.0 and min(nodes) > 9.0
    # the lowest request closes the run; it is the ladder's own end point, so the evolution up to
    # it went through nodes only
    assert allz[-1] == 9.0
    # no evolution: just the requests
    allz, nodes = D.required_redshifts(D.Inputs(SOURCE_MODEL=1), [8.0, 12.0, 8.0])
    assert allz == [12.0, 8.0] and nodes == {12.0, 8.0}
    # IONISE_ENTIRE_SPHERE keeps L_FACTOR x pixel as the smallest radius (IonisationBox.c:968-972)
    so, ap = D.S.default_simulation_options(HII_DIM=64, BOX_LEN=32.0), D.S.default_astro_params()
    assert D.ionisation_radii(so, ap, True, True) == D.ionisation_radii(so, ap, False)
    assert D.ionisation_radii(so, ap, True, False) <= D.ionisation_radii(so, ap, True, True)
