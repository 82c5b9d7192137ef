"""Shared inputs of the recombination-model tests (CPU oracle and HIP parity)."""

from __future__ import annotations

import importlib

import numpy as np
from scipy.interpolate import CubicSpline

S = importlib.import_module("21cmfast_amd.structs")
W = importlib.import_module("21cmfast_amd.workloads")

LNG_MIN, DZ = -10.0, float(np.float32(0.2))


def ln_gamma_knots():
    """recombinations.c:103: RR_lnGamma_min + gamma_ct * RR_DEL_lnGamma with a FLOAT step, i.e. the
    product is rounded to float before it is added."""
    g = np.arange(S.RR_NGAMMA, dtype=np.float32)
    return LNG_MIN + (g * np.float32(0.1)).astype(np.float64)



def synthetic_rr_tables():
    """A smooth stand-in for init_MHR's table (the grid algorithm only interpolates it): rows in
    z_ct, columns in ln Gamma_12, plus the natural-spline c coefficients (y''/2) of each row as
    gsl_interp_cspline holds them (reference: recombinations.c:94-122)."""
    lnG = ln_gamma_knots()
    z = (np.arange(S.RR_NZ) * DZ)[:, None]
    G = np.exp(lnG)[None, :]
    y = 0.04 * ((1 + z) / 9.0) ** 2.2 * G**0.45 / (1 + 0.3 * G**0.5) * (1 + 0.1 * np.sin(lnG))
    c = np.empty_like(y)
    for i in range(S.RR_NZ):
        c[i] = CubicSpline(lnG, y[i], bc_type="natural")(lnG, 2) / 2.0
    return np.ascontiguousarray(y), np.ascontiguousarray(c)


def recomb_spec(n, model=2, cell_recomb=1, lagrangian=True, hii_dim_z=None, ts=0,
                r_bubble_max=10.0):
    """IonizeSpec of the parity workloads with a recombination model switched on."""
    mode = W.FCOLL_STARS if lagrangian else W.FCOLL_ERFC
    spec = W.ionize_spec(n, mode=mode, r_bubble_max=r_bubble_max, hii_dim_z=hii_dim_z,
                         use_ts_fluct=ts)
    if not lagrangian:
        spec.hii_filter = 0      # recombination runs of the reference use the real-space top-hat
    y, c = synthetic_rr_tables()
    spec.recomb_model = model
    spec.cell_recomb = cell_recomb
    spec.rr_y = y.ctypes.data_as(S.c_double_p)
    spec.rr_c = c.ctypes.data_as(S.c_double_p)
    spec.gamma_prefactor = 2.5e-3 if lagrangian else 40.0
    spec.fabs_dtdz = 0.55   # |dt/dz| / 1e15 s at z ~ 9
    spec.dz = 0.2
    spec.first_snapshot = 0
    spec._rr = (y, c)  # keep alive
    return spec


def inputs(shape, seed=4, ts=False):
    rng = np.random.default_rng(seed)
    density = W.density_field_numpy(shape, seed=seed)
    n_ion = W.nion_from_density(density)
    out = {
        "density": density, "n_ion": n_ion,
        "whalo_sfr": np.ascontiguousarray(n_ion * (0.8 + 0.4 * rng.random(density.shape)) * 1e-9,
                                          np.float32),
        "prev_nrec": np.ascontiguousarray(0.6 * rng.random(density.shape) ** 2, np.float32),
        "prev_z_reion": np.where(rng.random(density.shape) < 0.1, 11.5, -1.0).astype(np.float32),
    }
    if ts:
        out["xe"] = (-0.05 + 0.5 * rng.random(density.shape) ** 3).astype(np.float32)
        out["Tneutral"] = (8.0 + 4.0 * rng.random(density.shape)).astype(np.float32)
    return out
