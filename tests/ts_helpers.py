"""Shared inputs of the spin-temperature tests (CPU oracle and HIP parity).

The cell algorithm only sees scalars and tables, so the parity workloads use real cosmology /
z' constants (oracle/ref_heating.py) with synthetic, smooth source grids and frequency tables
scaled to physical magnitudes: a few K of X-ray heating per step, x_e from 1e-4 to 0.3 (every
row of the x_e interpolation is visited), Lyman-alpha fluxes around 1e-11..1e-9 so that the
Wouthuysen-Field iteration runs, and a few cells below the 1e-20 flux switch."""

from __future__ import annotations

import ctypes as C
import importlib
import math

import numpy as np

S = importlib.import_module("21cmfast_amd.structs")

from oracle import ref_heating as RH  # noqa: E402
from oracle.ref_scalars import Cosmo  # noqa: E402

NX = S.X_INT_NXHII


def smooth_field(shape, rng, sigma, corr=3):
    """A cheap correlated Gaussian field: white noise box-averaged along every axis."""
    a = rng.standard_normal(shape)
    for ax in range(a.ndim):
        acc = np.zeros_like(a)
        for s in range(corr):
            acc += np.roll(a, s, axis=ax)
        a = acc / math.sqrt(corr)
    return (sigma * a / a.std()).astype(np.float32)


def lya_tables(rng):
    """Synthetic heating efficiencies [erg], smooth with a sign change like the real tables."""
    t = np.linspace(-1, 3, S.LYA_NT)
    g = np.linspace(1, 7, S.LYA_NGP)
    tk, ts, tg = np.meshgrid(t, t, g, indexing="ij")
    dEC = 2e-14 * (0.3 + np.tanh(1.5 - tk)) * (1 + 0.2 * np.sin(ts)) * (tg / 7.0)
    dEI = -1.2e-14 * (1 / (1 + np.exp(tk - 1.0))) * (1 + 0.1 * np.cos(2 * ts)) * (0.5 + tg / 14.0)
    return np.ascontiguousarray(dEC), np.ascontiguousarray(dEI)


def make(n=24, n_step=12, lagrangian=True, seed=5, zp=12.0, dzp=-0.3, lya_heating=True,
         xray_heating=True, cmb_heating=True, no_light=False, hii_dim_z=None, fcoll_tables=False):
    """Returns (spec, inputs dict).  inputs: density, previous (dict of three boxes), source
    (dict) or filtered_density."""
    rng = np.random.default_rng(seed)
    nz = hii_dim_z or n
    shape = (n, n, nz)
    c = Cosmo()
    spec = S.TsSpec(hii_dim=n, hii_dim_z=nz, n_step=n_step,
                    source_mode=S.TS_SRC_GRIDS if lagrangian else (
                        S.TS_SRC_FCOLL_TABLES if fcoll_tables else S.TS_SRC_SFRD_TABLE),
                    use_xray_heating=int(xray_heating), use_cmb_heating=int(cmb_heating),
                    use_lya_heating=int(lya_heating), no_light=int(no_light))
    spec.redshift = float(np.float32(zp))
    spec.dzp = float(np.float32(dzp))
    spec.growth_ratio = 1.0
    spec.clumping_factor = 2.0
    for k, v in RH.zp_consts(c, zp, lagrangian=lagrangian).items():
        setattr(spec, k, v)
    ze = RH.z_edges(c, zp, 64, 96.0, n_step=n_step, R_MAX_TS=300.0)
    keep = {}

    density = smooth_field(shape, rng, 0.35)
    density = np.maximum(density, -0.95).astype(np.float32)
    density.flat[3] = -1.3  # below -1: clamped to -1 + 1e-7 (:1800-1802)
    density.flat[7] = 0.0   # |delta| <= 1e-7: no structure-formation term (:1247)
    lnxe = rng.uniform(math.log(8e-5), math.log(0.3), shape)
    prev = {"xray_ionised_fraction": np.exp(lnxe).astype(np.float32),
            "kinetic_temp_neutral": np.exp(rng.uniform(math.log(4.0), math.log(800.0), shape)).astype(np.float32),
            "spin_temperature": np.exp(rng.uniform(math.log(8.0), math.log(300.0), shape)).astype(np.float32)}
    prev["xray_ionised_fraction"].flat[11] = 0.9995  # above the last table row (* 0.999 clamp)
    prev["kinetic_temp_neutral"].flat[13] = 6e4      # above MAX_TK: temperature frozen
    prev["kinetic_temp_neutral"].flat[17] = 0.6      # ln T < 0: first knot of the kappa tables

    # per-shell factors
    for i in range(n_step):
        zpp = ze["zpp"][i]
        if lagrangian:
            spec.z_edge_factor[i] = abs(ze["dzpp"][i] * ze["dtdz"][i])
        else:
            spec.z_edge_factor[i] = abs(ze["dzpp"][i] * ze["dtdz"][i]) * c.hubble(zpp) / 0.5
        spec.xray_R_factor[i] = (1 + zpp) ** -1.0
        spec.starlya_prefactor[i] = 1e-7 * (1 + 0.3 * math.cos(i)) * (i < n_step - 2)
        spec.lya_cont_prefactor[i] = 0.6 * spec.starlya_prefactor[i]
        spec.lya_inj_prefactor[i] = 0.4 * spec.starlya_prefactor[i]
        spec.zpp_growth[i] = ze["growth"][i]

    # frequency integrals: smooth in x_e and in the shell index; ion ~ 1, heat ~ 5e-12 erg, lya
    xs = np.log10(np.asarray(S.X_INT_XHII))[:, None]
    rs = np.arange(n_step)[None, :] / n_step
    fion = (1.5 - 0.2 * xs) * np.exp(-1.5 * rs) * (1 + 0.1 * np.sin(3 * xs + 5 * rs))
    fheat = 5e-12 * (0.2 + 0.1 * (xs + 4)) * np.exp(-1.2 * rs)
    flya = 3e6 * (0.6 - 0.1 * xs) * np.exp(-1.4 * rs)
    keep["freq"] = [np.ascontiguousarray(a, np.float64) for a in (fheat, fion, flya)]
    spec.freq_int_heat, spec.freq_int_ion, spec.freq_int_lya = (
        a.ctypes.data_as(S.c_double_p) for a in keep["freq"])
    if lya_heating:
        keep["lya"] = lya_tables(rng)
        spec.lya_dEC, spec.lya_dEI = (a.ctypes.data_as(S.c_double_p) for a in keep["lya"])

    inputs = {"density": density, "previous": prev, "source": None, "filtered_density": None}
    if lagrangian:
        sfr = np.empty((n_step,) + shape, np.float32)
        xray = np.empty((n_step,) + shape, np.float32)
        for i in range(n_step):
            f = np.exp(smooth_field(shape, rng, 0.8 / (1 + 0.3 * i)))
            sfr[i] = f * 1e-2
            xray[i] = f * (1 + 0.2 * rng.random(shape)) * 1e-2
        sfr[:, 0, 0, 5] = 0.0  # a cell no shell reaches: J_alpha from X-rays only
        xray[:, 0, 0, 5] = 0.0  # ... and none at all: the collisions-only branch
        inputs["source"] = {"filtered_sfr": sfr, "filtered_xray": xray}
        xray_terms = sum(float(xray[i].mean()) * spec.z_edge_factor[i] * spec.xray_R_factor[i] * 1e38
                         * fion[7, i] for i in range(n_step))
        lya_terms = sum(float(sfr[i].mean()) * spec.z_edge_factor[i] * spec.starlya_prefactor[i]
                        for i in range(n_step))
    else:
        fd = np.empty((n_step,) + shape, np.float32)
        # one spare row: upstream's lookup touches y[idx + 1] with weight 0 on the last knot
        tabs = np.zeros((n_step + 1, S.NDELTA_TABLE), np.float32)
        xray_terms = lya_terms = 0.0
        spec.sfr_scale = 0.05
        spec.xray_scale = 1e40 * RH.PC["s_per_yr"]
        for i in range(n_step):
            fd[i] = smooth_field(shape, rng, 2.5 / (1 + 0.25 * i))  # extrapolated to z = 0
            g = spec.zpp_growth[i]
            lo, hi = float(fd[i].min()) * g, float(fd[i].max()) * g * 1.001
            spec.tab_min[i] = lo
            spec.tab_width[i] = (hi - lo) / (S.NDELTA_TABLE - 1.0)
            x = lo + np.arange(S.NDELTA_TABLE) * spec.tab_width[i]
            tabs[i] = np.maximum(-9.0 + 4.0 * x - 0.5 * x * x - 0.1 * i, -50.0)
            if fcoll_tables:  # CONST-ION-EFF: linear tables, f_coll and (here) a multiple of it
                tabs[i] = np.exp(tabs[i])
                fc = np.interp(fd[i].astype(np.float64) * g, x, tabs[i].astype(np.float64))
            else:
                fc = np.exp(np.interp(fd[i].astype(np.float64) * g, x, tabs[i].astype(np.float64)))
            spec.mean_sfr_zpp[i] = 1.1 * fc.mean() * (1 + 0.05 * math.sin(i))
            sfr_mean = float(((1 + fd[i] * g) * fc).mean()) * 1.1 * (1 + 0.05 * math.sin(i)) * spec.sfr_scale
            xray_terms += (sfr_mean * spec.z_edge_factor[i] * spec.xray_scale * spec.xray_R_factor[i]
                           * fion[7, i])
            lya_terms += sfr_mean * spec.z_edge_factor[i] * spec.starlya_prefactor[i]
        keep["tabs"] = np.ascontiguousarray(tabs)
        if fcoll_tables:
            keep["dtabs"] = np.ascontiguousarray(
                tabs * (1.0 + 0.1 * np.cos(np.arange(n_step + 1))[:, None]), np.float32)
            spec.fcoll_tables = keep["tabs"].ctypes.data_as(S.c_float_p)
            spec.dfcoll_tables = keep["dtabs"].ctypes.data_as(S.c_float_p)
        else:
            spec.ln_sfrd_tables = keep["tabs"].ctypes.data_as(S.c_float_p)
        inputs["filtered_density"] = fd
    # scale the two radiative prefactors to physical magnitudes (see the module docstring)
    target_xion = 2.5e-4 / (abs(spec.dzp) * abs(spec.dt_dzp))  # delta x_e ~ 2.5e-4 per step
    spec.xray_prefactor = target_xion / (xray_terms * spec.volunit_inv)
    spec.lya_star_prefactor = 2e-10 / (lya_terms * spec.volunit_inv)
    spec._keep = keep
    return spec, inputs


def first_spec(n=24, z=30.0, hii_dim_z=None):
    c = Cosmo()
    d = RH.densities(c)
    return S.TsFirstSpec(hii_dim=n, hii_dim_z=hii_dim_z or n, redshift=float(np.float32(z)),
                         perturbed_redshift=float(np.float32(z)),
                         inverse_growth_factor_z=1 / c.dicke(z), growth_factor_zp=c.dicke(z),
                         xe=2.1e-4, TK=18.3, cT_ad=0.58 - 0.006 * (z - 10.0), No=d.No, N_b0=d.N_b0,
                         A10=RH.PC["A10"], T_21=RH.PC["T_21"], T_cmb=RH.PC["T_cmb"])


def c_array(values, ctype=C.c_double):
    return (ctype * len(values))(*values)


def add_minis(spec, inputs, seed=11, strength=1.0):
    """Switch the molecularly cooled population on for an SFRD_TABLE workload of make(): smooth
    2-D ln SFRD tables (overdensity x log10 M_crit,LW), shell-filtered turnover grids with
    structure, Pop-III Lyman-alpha / Lyman-Werner prefactors.  strength = 0 keeps the tables but
    gives the population no light (the run must then reproduce the one-population result)."""
    assert spec.source_mode == S.TS_SRC_SFRD_TABLE
    rng = np.random.default_rng(seed)
    n_step = spec.n_step
    shape = inputs["density"].shape
    fd = inputs["filtered_density"]
    nd, nm = S.NDELTA_TABLE, S.NMTURN_TABLE
    spec.use_mini_halos = 1
    spec.mturn_tab_min = 5.0 - 9e-8
    spec.mturn_tab_width = (10.0 - spec.mturn_tab_min) / (nm - 1.0)
    spec.sfr_scale_mini = 0.01 * strength
    spec.xray_scale_mini = 3e40 * RH.PC["s_per_yr"]
    mcrit = np.empty((n_step,) + shape, np.float32)
    tabs = np.zeros((n_step + 1, nd, nm), np.float32)  # one spare table like the 1-D case
    y = (spec.mturn_tab_min + spec.mturn_tab_width * np.arange(nm))[None, :]
    for i in range(n_step):
        mcrit[i] = (6.0 + 0.5 * np.tanh(smooth_field(shape, rng, 1.0)) / (1 + 0.1 * i)
                    + 0.02 * i).astype(np.float32)
        x = (spec.tab_min[i] + np.arange(nd) * spec.tab_width[i])[:, None]
        tabs[i] = np.maximum(-10.0 + 3.0 * x - 0.4 * x * x - 0.1 * i - 1.2 * (y - 6.0), -50.0)
        g = spec.zpp_growth[i]
        d = fd[i].astype(np.float64) * g
        fc = np.exp(-10.0 + 3.0 * d - 0.4 * d * d - 0.1 * i - 1.2 * (mcrit[i].astype(np.float64) - 6.0))
        spec.mean_sfr_zpp_mini[i] = 0.9 * fc.mean() * (1 + 0.05 * math.cos(i))
        spec.starlya_prefactor_mini[i] = 0.7 * spec.starlya_prefactor[i]
        spec.lya_cont_prefactor_mini[i] = 0.5 * spec.starlya_prefactor_mini[i]
        spec.lya_inj_prefactor_mini[i] = 0.5 * spec.starlya_prefactor_mini[i]
        spec.lw_prefactor[i] = 2.5e7 * (1 + 0.2 * math.sin(i)) * (i < n_step - 3)
        spec.lw_prefactor_mini[i] = 1.8 * spec.lw_prefactor[i]
    spec._keep["tabs_mini"] = np.ascontiguousarray(tabs)
    spec._keep["mcrit"] = mcrit
    spec.ln_sfrd_tables_mini = spec._keep["tabs_mini"].ctypes.data_as(S.c_float_p)
    spec.filtered_log10_mcrit = mcrit.ctypes.data_as(S.c_float_p)
    inputs["filtered_log10_mcrit"] = mcrit
    return spec, inputs


def add_minis_grids(spec, inputs, seed=13, lw_copies=False):
    """The molecularly cooled population for a source-grid (Lagrangian) workload of make():
    a filtered_sfr_mini grid per shell, Pop-III prefactors, optionally the straight-line copies
    the Lyman-Werner sums read under LYA_MULTIPLE_SCATTERING."""
    assert spec.source_mode == S.TS_SRC_GRIDS
    rng = np.random.default_rng(seed)
    n_step = spec.n_step
    src = inputs["source"]
    shape = src["filtered_sfr"].shape[1:]
    spec.use_mini_halos = 1
    mini = np.empty_like(src["filtered_sfr"])
    for i in range(n_step):
        mini[i] = 3e-3 * np.exp(smooth_field(shape, rng, 0.6 / (1 + 0.3 * i)))
        spec.starlya_prefactor_mini[i] = 0.7 * spec.starlya_prefactor[i]
        spec.lya_cont_prefactor_mini[i] = 0.5 * spec.starlya_prefactor_mini[i]
        spec.lya_inj_prefactor_mini[i] = 0.5 * spec.starlya_prefactor_mini[i]
        spec.lw_prefactor[i] = 2.5e7 * (1 + 0.2 * math.sin(i)) * (i < n_step - 3)
        spec.lw_prefactor_mini[i] = 1.8 * spec.lw_prefactor[i]
    src["filtered_sfr_mini"] = mini
    if lw_copies:
        src["filtered_sfr_lw"] = (src["filtered_sfr"] * (0.8 + 0.4 * rng.random(mini.shape))).astype(np.float32)
        src["filtered_sfr_mini_lw"] = (mini * (0.8 + 0.4 * rng.random(mini.shape))).astype(np.float32)
    return spec, inputs
