"""Pins the oracle's ComputePerturbedField restatement with the reference's own
known-answer test, restated from /root/reference/tests/test_perturb.py:52-135:

fake ICs whose (low-res) velocities displace every particle by exactly one HII_DIM cell
(+1 in y by the first-order term, -1 in z by the 2LPT term) must yield the IC low-res
density rolled by that many cells and multiplied by the growth factor of the INITIAL
redshift, to atol 1e-3.  Same geometry: HII_DIM=4, DIM=12, BOX_LEN=8.
"""

import importlib

import numpy as np
import pytest

S = importlib.import_module("21cmfast_amd.structs")

HII_DIM, DIM, BOX_LEN = 4, 12, 8.0
D_Z, D_ZI = 0.1411, 0.004234  # ~ D(z=8), D(z=300); any pair works, they are explicit inputs


def perturb_spec(algorithm, **kw):
    spec = S.PerturbSpec(
        dim=DIM, dim_z=DIM, hii_dim=HII_DIM, hii_dim_z=HII_DIM, box_len=BOX_LEN, box_len_z=BOX_LEN,
        perturb_algorithm=algorithm, perturb_on_high_res=0, keep_3d_velocities=0,
        smooth_evolved_density=0, density_smooth_radius_mpc=0.2 * BOX_LEN / HII_DIM,
        growth_factor=D_Z, init_growth_factor=D_ZI, dDdt_over_D=1.3e-16,
    )
    return spec.update(**kw)


def fake_ics(algorithm):
    """test_perturb.py:52-106 (get_fake_ics), low-res perturbation branch."""
    res_fac = DIM // HII_DIM
    cell = BOX_LEN / HII_DIM
    fac_1lpt = cell / (D_Z - D_ZI)
    fac_2lpt = cell / ((-3.0 / 7.0) * (D_Z**2 - D_ZI**2))
    lo, hi = (HII_DIM,) * 3, (DIM,) * 3
    ics = {k: np.zeros(lo, np.float32) for k in
           ("lowres_density", "lowres_vx", "lowres_vy", "lowres_vz", "lowres_vx_2LPT",
            "lowres_vy_2LPT", "lowres_vz_2LPT")}
    ics["lowres_vy"][...] = fac_1lpt
    if algorithm == 2:
        ics["lowres_vz_2LPT"][...] = fac_2lpt
    ics["lowres_density"][0, 0, 0] = 1
    ics["lowres_density"][HII_DIM // 2, HII_DIM // 2, HII_DIM // 2] = -1
    d_hi = np.zeros(hi, np.float32)
    d_hi[0, 0, 0] = res_fac**3
    d_hi[DIM // 2, DIM // 2, DIM // 2] = -(res_fac**3)
    ics["hires_density"] = d_hi
    return ics


def expected_density(ics, algorithm):
    roll = {0: (0, 0, 0), 1: (0, 1, 0), 2: (0, 1, -1)}[algorithm]
    d = D_Z if algorithm == 0 else D_ZI
    return np.roll(ics["lowres_density"], roll, (0, 1, 2)) * d


@pytest.mark.parametrize("algorithm", [2, 1, 0])  # 2LPT, ZELDOVICH, LINEAR
def test_lowres_perturb_known_answer(oracle, algorithm):
    ics = fake_ics(algorithm)
    out = oracle.perturb_grids(perturb_spec(algorithm), ics)
    np.testing.assert_allclose(out["density"], expected_density(ics, algorithm), atol=1e-3)


def test_mass_conservation_and_clip(oracle):
    """CIC deposits conserve mass: mean(delta) = 0 for zero-mean ICs; delta >= -1 + 1e-7."""
    rng = np.random.default_rng(0)
    n, N = 8, 16
    ics = {k: (0.5 * rng.standard_normal((n,) * 3)).astype(np.float32) for k in
           ("lowres_vx", "lowres_vy", "lowres_vz", "lowres_vx_2LPT", "lowres_vy_2LPT",
            "lowres_vz_2LPT")}
    d = rng.standard_normal((N,) * 3).astype(np.float32)
    ics["hires_density"] = d - d.mean()
    ics["lowres_density"] = np.zeros((n,) * 3, np.float32)
    spec = perturb_spec(2, dim=N, dim_z=N, hii_dim=n, hii_dim_z=n, box_len=12.0, box_len_z=12.0,
                        growth_factor=0.9, init_growth_factor=0.05, keep_3d_velocities=1)
    out = oracle.perturb_grids(spec, ics)
    assert abs(out["density"].astype(np.float64).mean()) < 2e-6
    assert out["density"].min() >= np.float32(-1 + 1e-7)
    # velocity of the DC mode is removed: each component has zero mean
    for ax in "xyz":
        assert abs(out["velocity_" + ax].astype(np.float64).mean()) < 1e-9
