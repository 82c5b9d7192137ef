"""Synthetic inputs of the mini-halo (USE_MINI_HALOS, E-INTEGRAL) ionisation path shared by the
oracle and GPU tests: smooth 2-D ln N_ion tables (overdensity x log10 turnover mass), turnover-mass
grids with structure, a previous snapshot's density and per-radius f_coll history."""
import importlib

import numpy as np

S = importlib.import_module("21cmfast_amd.structs")
W = importlib.import_module("21cmfast_amd.workloads")

_KEEP = []  # ctypes callbacks must outlive the specs that point at them


def ln_f_acg(delta, l10mt, r_index, prev):
    scale = 0.7 if prev else 1.0
    return (np.log(scale * 0.04 * (1 + np.maximum(delta, -0.999)) ** 1.5 / (1 + 0.05 * r_index))
            - 0.3 * (l10mt - 8.0))


def ln_f_mcg(delta, l10mt, r_index, prev):
    scale = 0.6 if prev else 1.0
    return (np.log(scale * 0.004 * (1 + np.maximum(delta, -0.999)) ** 1.2 / (1 + 0.05 * r_index))
            - 0.5 * (l10mt - 6.0))


def install_tables2d(spec, calls=None):
    nd, nm = S.NDELTA_TABLE, S.NMTURN_TABLE

    def table2d_fn(r_index, prev, dmin, dmax, amin, amax, mmin, mmax, tab_a, tab_m, user):
        if calls is not None:
            calls.append((r_index, prev, dmin, dmax, amin, amax, mmin, mmax))
        x = (dmin + (dmax - dmin) / (nd - 1.0) * np.arange(nd))[:, None]
        ya = (amin + (amax - amin) / (nm - 1.0) * np.arange(nm))[None, :]
        ym = (mmin + (mmax - mmin) / (nm - 1.0) * np.arange(nm))[None, :]
        a = np.ascontiguousarray(ln_f_acg(x, ya, r_index, prev), np.float32).ravel()
        m = np.ascontiguousarray(ln_f_mcg(x, ym, r_index, prev), np.float32).ravel()
        np.ctypeslib.as_array(tab_a, (nd * nm,))[:] = a
        np.ctypeslib.as_array(tab_m, (nd * nm,))[:] = m
        return 0

    cb = S.TABLE2D_FN(table2d_fn)
    _KEEP.append(cb)
    spec.table2d_fn = cb
    return spec


def mini_spec(n, nz=None, need_prev=1, zeta_mini=60.0, recomb_model=0, **kw):
    spec = W.ionize_spec(n, hii_dim_z=nz, mode=W.FCOLL_TABLE_EXP, **kw)
    spec.fix_mean = 0       # Sheth-Tormen runs do not rescale to the global mean
    spec.mass_dep_zeta = 1
    if recomb_model:
        from recomb_helpers import synthetic_rr_tables
        y, c = synthetic_rr_tables()
        spec.recomb_model = recomb_model
        spec.rr_y = y.ctypes.data_as(S.c_double_p)
        spec.rr_c = c.ctypes.data_as(S.c_double_p)
        spec.gamma_prefactor = 40.0
        spec.fabs_dtdz = 0.55
        spec.dz = 0.2
        spec.first_snapshot = 0
        spec._rr = (y, c)
    spec.use_mini_halos = 1
    spec.need_prev_ion = need_prev
    spec.ion_eff_factor_mini = zeta_mini
    spec.mean_f_coll_mini = 0.004
    spec.f_limit_mcg = 1e-7
    spec.gamma_prefactor_mini = spec.gamma_prefactor * zeta_mini / spec.ion_eff_factor
    return install_tables2d(spec)


def mini_inputs(shape, n_radii, seed=3, history=True):
    """prev_density, the two log10 M_turn grids [N], previous f_coll history [n_radii, N]"""
    rng = np.random.default_rng(seed)
    density = W.density_field_numpy(shape, seed=seed + 100)
    prev_density = (0.8 * density + 0.02 * rng.standard_normal(shape)).astype(np.float32)
    prev_density = np.maximum(prev_density, -0.98).astype(np.float32)
    # turnover masses: a floor plus patches raised by feedback (cells "ionised earlier")
    patch = W.density_field_numpy(shape, seed=seed + 7) > 0.15
    mt_a = np.where(patch, 9.1 + 0.2 * rng.random(shape), 8.3).astype(np.float32)
    mt_m = np.where(patch, 9.1 + 0.2 * rng.random(shape), 6.2 + 0.8 * rng.random(shape))
    mt_m = mt_m.astype(np.float32)
    if history:
        prev_nion = (0.02 * rng.random((n_radii,) + tuple(shape))).astype(np.float32)
        prev_mini = (0.004 * rng.random((n_radii,) + tuple(shape))).astype(np.float32)
    else:
        prev_nion = np.zeros((n_radii,) + tuple(shape), np.float32)
        prev_mini = np.zeros((n_radii,) + tuple(shape), np.float32)
    return density, dict(prev_density=prev_density, log10_mturn_acg=mt_a, log10_mturn_mcg=mt_m,
                         prev_nion=prev_nion, prev_nion_mini=prev_mini)
