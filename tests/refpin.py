"""Shared helpers of the reference-fixture pins (tests/test_reference_fixtures.py on the CPU
oracle, tests/test_gpu_reference_fixtures.py on the HIP path).

The configuration is the reference's integration-test default
(reference: tests/produce_integration_test_data.py:48-63): HII_DIM = 50, DIM = 150,
BOX_LEN = 100 Mpc, random_seed = 12345, Planck18 cosmology, 2LPT on the low-resolution grid,
INITIAL_REDSHIFT = 300; host scalars come from oracle/ref_scalars.py (numpy / scipy), i.e.
from neither the library under test nor the oracle's C code.
"""

from __future__ import annotations

import importlib
from pathlib import Path

import numpy as np

from oracle import ref_scalars as RS
from oracle.h5mini import H5File
from oracle.powerbox_power import get_power, pdf_histogram  # noqa: F401

S = importlib.import_module("21cmfast_amd.structs")

DATA = Path(__file__).resolve().parent / "golden" / "reference"
HII_DIM, DIM, BOX_LEN, SEED = 50, 150, 100.0, 12345
INITIAL_REDSHIFT, DENSITY_SMOOTH_RADIUS = 300.0, 0.2
VEL_NORM = 1e16  # produce_integration_test_data.py:377

# name -> (PERTURB_ALGORITHM, PERTURB_ON_HIGH_RES)   (OPTIONS_PT, :287-292)
PT_CASES = {"simple": (2, 0), "no2lpt": (1, 0), "linear": (0, 0), "highres": (2, 1)}

_cosmo = None


def cosmo() -> RS.Cosmo:
    global _cosmo
    if _cosmo is None:
        _cosmo = RS.Cosmo()
    return _cosmo


def fixture(kind: str, name: str) -> H5File:
    return H5File(DATA / f"{kind}_{name}.h5")


def volume_f32(box_len: float) -> float:
    """indexing.h VOLUME: the float product BOX_LEN * BOX_LEN * NON_CUBIC_FACTOR * BOX_LEN."""
    L = np.float32(box_len)
    return float(np.float32(np.float32(L * L) * np.float32(1.0)) * L)


def ics_spec(algorithm=2, hires=0, n_threads=2, rng_stream=1):
    pk = RS.pk_table(cosmo(), DIM, BOX_LEN)
    spec = S.IcsSpec(dim=DIM, dim_z=DIM, hii_dim=HII_DIM, hii_dim_z=HII_DIM, box_len=BOX_LEN,
                     box_len_z=BOX_LEN, volume=volume_f32(BOX_LEN), perturb_algorithm=algorithm,
                     perturb_on_high_res=hires, n_m=len(pk),
                     pk_by_m=pk.ctypes.data_as(S.c_double_p), seed=SEED, rng_stream=rng_stream,
                     rng_threads=n_threads)
    spec._pk = pk  # keep the table alive
    return spec


def perturb_spec(z: float, algorithm=2, hires=0):
    c = cosmo()
    return S.PerturbSpec(
        dim=DIM, dim_z=DIM, hii_dim=HII_DIM, hii_dim_z=HII_DIM, box_len=BOX_LEN,
        box_len_z=BOX_LEN, perturb_algorithm=algorithm, perturb_on_high_res=hires,
        keep_3d_velocities=0, smooth_evolved_density=0,
        density_smooth_radius_mpc=DENSITY_SMOOTH_RADIUS * BOX_LEN / HII_DIM,
        growth_factor=c.dicke(z), init_growth_factor=c.dicke(INITIAL_REDSHIFT),
        dDdt_over_D=c.ddickedt(z) / c.dicke(z))


def check_perturb_fixture(name: str, density: np.ndarray, velocity_z: np.ndarray):
    """The four asserts of the reference's test_perturb_field_data
    (tests/test_integration_features.py:305-308), at ITS tolerances."""
    f = fixture("perturb_field_data", name)
    p_dens, k = get_power(density, BOX_LEN)
    p_vel, _ = get_power(velocity_z * VEL_NORM, BOX_LEN)
    _, y_dens = pdf_histogram(density, -0.8, 2.0, 50)
    _, y_vel = pdf_histogram(velocity_z * VEL_NORM, -2, 2, 50)
    np.testing.assert_allclose(k, f["k_dens"], rtol=1e-12)
    np.testing.assert_allclose(p_dens, f["power_dens"], atol=5e-3, rtol=1e-3)
    np.testing.assert_allclose(p_vel, f["power_vel"], atol=5e-3, rtol=1e-3)
    np.testing.assert_allclose(y_dens, f["pdf_dens"], atol=5e-3, rtol=1e-3)
    np.testing.assert_allclose(y_vel, f["pdf_vel"], atol=5e-3, rtol=1e-3)
    return float(np.abs(p_dens / f["power_dens"] - 1).max())


def check_coeval_fields(name: str, fields: dict, rtol: float = 1e-3):
    """Binned power of IC / PerturbedField outputs against a power_spectra_*.h5 fixture."""
    f = fixture("power_spectra", name)
    worst = {}
    for key, arr in fields.items():
        p, k = get_power(arr, BOX_LEN)
        ref = f[f"coeval/power_{key}"]
        np.testing.assert_allclose(k, f["coeval/k"], rtol=1e-12)
        np.testing.assert_allclose(p, ref, rtol=rtol, atol=0, err_msg=key)
        worst[key] = float(np.abs(p / ref - 1).max())
    return worst
