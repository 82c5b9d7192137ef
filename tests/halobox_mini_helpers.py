"""Synthetic mini-halo inputs of the HaloBox tests: smooth 2-D ln tables (overdensity x log10
turnover mass), turnover grids with structure, previous-box fields for get_log10_turnovers."""
import importlib
import math

import numpy as np

from test_oracle_halobox import halobox_spec, make_tables

S = importlib.import_module("21cmfast_amd.structs")


def add_minis(spec, n, seed=4, xray=True):
    tab_min, tab_width = spec.tab_min, spec.tab_width
    nd, nm = S.NDELTA_TABLE, S.NMTURN_TABLE
    rng = np.random.default_rng(seed)
    shape = (n, n, n)
    mta = (8.4 + 0.5 * rng.random(shape) ** 2).astype(np.float32)
    mtm = (5.6 + 1.5 * rng.random(shape)).astype(np.float32)
    spec.use_mini_halos = 1
    spec.mta_min, spec.mtm_min = float(mta.min()) * 0.999, float(mtm.min()) * 0.999
    spec.mta_width = (float(mta.max()) * 1.001 - spec.mta_min) / (nm - 1.0)
    spec.mtm_width = (float(mtm.max()) * 1.001 - spec.mtm_min) / (nm - 1.0)
    spec.mt_fixed_min = 5.0 - 9e-8
    spec.mt_fixed_width = (10.0 - spec.mt_fixed_min) / (nm - 1.0)
    x = (tab_min + tab_width * np.arange(nd))[:, None]
    ya = (spec.mta_min + spec.mta_width * np.arange(nm))[None, :]
    ym = (spec.mtm_min + spec.mtm_width * np.arange(nm))[None, :]
    yf = (spec.mt_fixed_min + spec.mt_fixed_width * np.arange(nm))[None, :]
    tabs = {"ln_nion_table2d": -9.0 + 4.5 * x - 0.7 * x * x - 0.8 * (ya - 8.5),
            "ln_nion_mini_table2d": -11.0 + 3.9 * x - 0.5 * x * x - 1.1 * (ym - 6.0),
            "ln_sfrd_mini_table2d": -8.0 + 3.5 * x - 0.4 * x * x - 1.0 * (yf - 6.0)}
    if xray:
        tabs["ln_xray_table2d"] = -2.0 + 4.1 * x - 0.6 * x * x - 0.3 * (yf - 6.0)
        spec.prefactor_xray = 4.4e3
    keep = [mta, mtm]
    for k, v in tabs.items():
        # one spare row: the lookups read [idx + 1] with weight 0 on the last knot
        a = np.zeros((nd + 1, nm), np.float32)
        a[:nd] = v
        keep.append(a)
        setattr(spec, k, a.ctypes.data_as(S.c_float_p))
    spec.log10_mturn_acg = mta.ctypes.data_as(S.c_float_p)
    spec.log10_mturn_mcg = mtm.ctypes.data_as(S.c_float_p)
    spec.prefactor_nion_mini, spec.prefactor_sfr_mini = 8.0e8, 5.0e-8
    spec._keep = tuple(spec._keep) + tuple(keep)
    spec._mt = (mta, mtm)
    return spec


def mini_spec(n, lpt2=1, xray=True):
    return add_minis(halobox_spec(n, n, False, make_tables(), lpt2=lpt2), n, xray=xray)


def turnover_inputs(shape, seed=6):
    rng = np.random.default_rng(seed)
    spec = S.MturnSpec(hii_dim=shape[0], hii_dim_z=shape[2], redshift=11.0, mturn_a_nofb=1.6e8,
                       vcb_const=18.0, A_LW=2.0, BETA_LW=0.6, A_VCB=1.0, BETA_VCB=1.8,
                       sigma_vcb=25.86 * math.sqrt(3 * math.pi / 8))
    g12 = (0.4 * rng.random(shape)).astype(np.float32)
    # few, scattered early-reionised cells: the running maximum shows between them
    zre = np.where(rng.random(shape) < 0.01, 12.0 + 5 * rng.random(shape), -1.0).astype(np.float32)
    j21 = (0.5 * rng.random(shape) ** 2).astype(np.float32)
    vcb = (30 * rng.random(shape)).astype(np.float32)
    return spec, g12, zre, j21, vcb
