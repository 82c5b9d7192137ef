"""ctypes mirrors of the C ABI structs in ``include/c21cm_abi.h`` / ``c21cm_grid.h``.

py21cmfast reaches the same structs through CFFI (``wrapper/structs.py:48-93`` builds
``ffi.new("struct X*")`` and fills it field by field); this module is the equivalent
for callers that do not have CFFI (tests, bench.py, the stand-alone host mirror in
``single_field.py``).  Field order and C types must stay in lock-step with the
headers; ``tests/test_abi_layout.py`` checks sizes and offsets against values the C
compiler reports.

Default values are the ones that arrive in the C structs after py21cmfast's
Python-side transforms (reference: src/py21cmfast/wrapper/inputs.py, SURVEY.md
Appendix A).
"""

from __future__ import annotations

import ctypes as C

MAX_RADII = 256
NDELTA_TABLE = 400

c_float_p = C.POINTER(C.c_float)
c_double_p = C.POINTER(C.c_double)


class _Base(C.Structure):
    """Structure with keyword construction + dict export."""

    def update(self, **kw):
        for k, v in kw.items():
            if not any(k == f[0] for f in self._fields_):
                raise AttributeError(f"{type(self).__name__} has no field {k!r}")
            setattr(self, k, v)
        return self

    def asdict(self):
        out = {}
        for name, typ in self._fields_:
            val = getattr(self, name)
            if isinstance(val, (int, float, bool)):
                out[name] = val
        return out


class CosmoParams(_Base):
    _fields_ = [
        ("hlittle", C.c_float),
        ("OMm", C.c_float),
        ("OMl", C.c_float),
        ("OMb", C.c_float),
        ("POWER_INDEX", C.c_float),
        ("OMn", C.c_float),
        ("OMk", C.c_float),
        ("OMr", C.c_float),
        ("OMtot", C.c_float),
        ("Y_He", C.c_float),
        ("wl", C.c_float),
    ]


class SimulationOptions(_Base):
    _fields_ = [
        ("HII_DIM", C.c_int),
        ("DIM", C.c_int),
        ("BOX_LEN", C.c_float),
        ("NON_CUBIC_FACTOR", C.c_float),
        ("N_THREADS", C.c_int),
        ("Z_HEAT_MAX", C.c_double),
        ("ZPRIME_STEP_FACTOR", C.c_double),
        ("SAMPLER_MIN_MASS", C.c_float),
        ("SAMPLER_BUFFER_FACTOR", C.c_double),
        ("N_COND_INTERP", C.c_int),
        ("N_PROB_INTERP", C.c_int),
        ("MIN_LOGPROB", C.c_double),
        ("HALOMASS_CORRECTION", C.c_double),
        ("PARKINSON_G0", C.c_double),
        ("PARKINSON_y1", C.c_double),
        ("PARKINSON_y2", C.c_double),
        ("INITIAL_REDSHIFT", C.c_float),
        ("DELTA_R_FACTOR", C.c_double),
        ("DENSITY_SMOOTH_RADIUS", C.c_double),
        ("DEXM_OPTIMIZE_MINMASS", C.c_double),
        ("DEXM_R_OVERLAP", C.c_double),
        ("CORR_STAR", C.c_double),
        ("CORR_SFR", C.c_double),
        ("CORR_LX", C.c_double),
        ("MIN_XE_FOR_FCOLL_IN_TAUX", C.c_double),
    ]


class MatterOptions(_Base):
    _fields_ = [
        ("USE_FFTW_WISDOM", C.c_bool),
        ("HMF", C.c_int),
        ("V_CB_MODEL", C.c_int),
        ("POWER_SPECTRUM", C.c_int),
        ("USE_INTERPOLATION_TABLES", C.c_int),
        ("PERTURB_ON_HIGH_RES", C.c_bool),
        ("PERTURB_ALGORITHM", C.c_int),
        ("MINIMIZE_MEMORY", C.c_bool),
        ("KEEP_3D_VELOCITIES", C.c_bool),
        ("DEXM_OPTIMIZE", C.c_bool),
        ("FILTER", C.c_int),
        ("HALO_FILTER", C.c_int),
        ("SMOOTH_EVOLVED_DENSITY_FIELD", C.c_bool),
        ("SOURCE_MODEL", C.c_int),
        ("SAMPLE_METHOD", C.c_int),
    ]


class AstroParams(_Base):
    _fields_ = [
        ("HII_EFF_FACTOR", C.c_float),
        ("F_STAR10", C.c_float),
        ("ALPHA_STAR", C.c_float),
        ("ALPHA_STAR_MINI", C.c_float),
        ("SIGMA_STAR", C.c_float),
        ("UPPER_STELLAR_TURNOVER_MASS", C.c_double),
        ("UPPER_STELLAR_TURNOVER_INDEX", C.c_double),
        ("F_STAR7_MINI", C.c_float),
        ("t_STAR", C.c_float),
        ("SIGMA_SFR_INDEX", C.c_double),
        ("SIGMA_SFR_LIM", C.c_double),
        ("L_X", C.c_double),
        ("L_X_MINI", C.c_double),
        ("SIGMA_LX", C.c_double),
        ("F_ESC10", C.c_float),
        ("ALPHA_ESC", C.c_float),
        ("F_ESC7_MINI", C.c_float),
        ("T_RE", C.c_float),
        ("M_TURN", C.c_float),
        ("R_BUBBLE_MAX", C.c_float),
        ("ION_Tvir_MIN", C.c_float),
        ("F_H2_SHIELD", C.c_double),
        ("NU_X_THRESH", C.c_float),
        ("X_RAY_SPEC_INDEX", C.c_float),
        ("X_RAY_Tvir_MIN", C.c_float),
        ("A_LW", C.c_double),
        ("BETA_LW", C.c_double),
        ("A_VCB", C.c_double),
        ("BETA_VCB", C.c_double),
        ("V_CB_AVG_DEBUG", C.c_double),
        ("POP2_ION", C.c_double),
        ("POP3_ION", C.c_double),
        ("PHOTONCONS_CALIBRATION_END", C.c_double),
        ("CLUMPING_FACTOR", C.c_double),
        ("ALPHA_UVB", C.c_double),
        ("R_MAX_TS", C.c_float),
        ("N_STEP_TS", C.c_int),
        ("DELTA_R_HII_FACTOR", C.c_double),
        ("R_BUBBLE_MIN", C.c_float),
        ("MAX_DVDR", C.c_double),
        ("NU_X_MAX", C.c_double),
        ("NU_X_BAND_MAX", C.c_double),
    ]


class AstroOptions(_Base):
    _fields_ = [
        ("USE_MINI_HALOS", C.c_bool),
        ("USE_X_RAY_HEATING", C.c_bool),
        ("USE_CMB_HEATING", C.c_bool),
        ("USE_LYA_HEATING", C.c_bool),
        ("RECOMB_MODEL", C.c_int),
        ("USE_TS_FLUCT", C.c_bool),
        ("M_MIN_in_Mass", C.c_bool),
        ("USE_EXP_FILTER", C.c_bool),
        ("CELL_RECOMB", C.c_bool),
        ("LYA_MULTIPLE_SCATTERING", C.c_bool),
        ("USE_ADIABATIC_FLUCTUATIONS", C.c_bool),
        ("PHOTON_CONS_TYPE", C.c_int),
        ("USE_UPPER_STELLAR_TURNOVER", C.c_bool),
        ("HALO_SCALING_RELATIONS_MEDIAN", C.c_bool),
        ("HII_FILTER", C.c_int),
        ("HEAT_FILTER", C.c_int),
        ("IONISE_ENTIRE_SPHERE", C.c_bool),
        ("INTEGRATION_METHOD_ATOMIC", C.c_int),
        ("INTEGRATION_METHOD_MINI", C.c_int),
    ]


class Table1D(_Base):
    _fields_ = [("size", C.c_int), ("x_values", c_double_p), ("y_values", c_double_p)]


class CosmoTables(_Base):
    _fields_ = [
        ("transfer_density", C.POINTER(Table1D)),
        ("transfer_vcb", C.POINTER(Table1D)),
        ("ps_norm", C.c_double),
        ("USE_SIGMA_8", C.c_bool),
        ("V_CB_AVG", C.c_double),
    ]


class ConfigSettings(_Base):
    _fields_ = [
        ("HALO_CATALOG_MEM_FACTOR", C.c_double),
        ("EXTRA_HALOBOX_FIELDS", C.c_bool),
        ("external_table_path", C.c_char_p),
        ("wisdoms_path", C.c_char_p),
    ]


class InitialConditionsStruct(_Base):
    _fields_ = [
        (n, c_float_p)
        for n in (
            "lowres_density", "lowres_vx", "lowres_vy", "lowres_vz",
            "lowres_vx_2LPT", "lowres_vy_2LPT", "lowres_vz_2LPT",
            "hires_density", "hires_vx", "hires_vy", "hires_vz",
            "hires_vx_2LPT", "hires_vy_2LPT", "hires_vz_2LPT",
            "lowres_vcb",
        )
    ]


class PerturbedFieldStruct(_Base):
    _fields_ = [(n, c_float_p) for n in ("density", "velocity_x", "velocity_y", "velocity_z")]


class HaloBoxStruct(_Base):
    _fields_ = [
        (n, c_float_p)
        for n in (
            "halo_mass", "halo_stars", "halo_stars_mini", "count", "n_ion", "halo_sfr",
            "halo_xray", "halo_sfr_mini", "whalo_sfr",
        )
    ] + [("log10_Mcrit_ACG_ave", C.c_double), ("log10_Mcrit_MCG_ave", C.c_double)]


class TsBoxStruct(_Base):
    _fields_ = [
        (n, c_float_p)
        for n in ("spin_temperature", "xray_ionised_fraction", "kinetic_temp_neutral", "J_21_LW")
    ] + [("Q_HI", C.c_double)]


class IonizedBoxStruct(_Base):
    _fields_ = [
        ("mean_f_coll", C.c_double),
        ("mean_f_coll_MINI", C.c_double),
        ("log10_Mturnover_ave", C.c_double),
        ("log10_Mturnover_MINI_ave", C.c_double),
    ] + [
        (n, c_float_p)
        for n in (
            "neutral_fraction", "ionisation_rate_G12", "mean_free_path", "z_reion",
            "cumulative_recombinations", "kinetic_temperature", "unnormalised_nion",
            "unnormalised_nion_mini",
        )
    ]


class HaloCatalogStruct(_Base):
    """``HaloCatalog`` (include/c21cm_abi.h; reference _outputstructs_wrapper.h:18-28)."""

    _fields_ = [
        ("n_halos", C.c_ulonglong), ("buffer_size", C.c_ulonglong),
        ("halo_masses", c_float_p), ("halo_coords", c_float_p),
        ("star_rng", c_float_p), ("sfr_rng", c_float_p), ("xray_rng", c_float_p),
    ]


def halo_catalog(masses, coords, star_rng, sfr_rng, xray_rng=None) -> "HaloCatalogStruct":
    """A HaloCatalog over float32 numpy arrays (masses [n], coords [n, 3] in Mpc, the standard-normal
    deviates of the three scaling relations); the arrays are kept alive on the struct."""
    import numpy as np

    arrs = [np.ascontiguousarray(a, np.float32) if a is not None else None
            for a in (masses, coords, star_rng, sfr_rng, xray_rng)]
    n = arrs[0].size
    if arrs[1].size != 3 * n or any(a is not None and a.size != n for a in arrs[2:]):
        raise ValueError("halo catalogue arrays of different lengths")
    cat = HaloCatalogStruct(n_halos=n, buffer_size=n)
    for name, a in zip(("halo_masses", "halo_coords", "star_rng", "sfr_rng", "xray_rng"), arrs):
        if a is not None:
            setattr(cat, name, a.ctypes.data_as(c_float_p))
    cat._keep = arrs
    return cat


class HaloConsts(_Base):
    """``c21cm_halo_consts`` (include/c21cm_grid.h)."""

    _fields_ = [(k, C.c_double) for k in (
        "redshift", "fstar_10", "alpha_star", "sigma_star", "alpha_upper", "pivot_upper",
        "upper_pivot_ratio", "fstar_7", "alpha_star_mini", "acg_thresh", "baryon_ratio", "t_h", "t_star",
        "sigma_sfr_lim", "sigma_sfr_idx", "l_x", "l_x_mini", "sigma_xray", "fesc_10", "fesc_7",
        "alpha_esc", "pop2_ion", "pop3_ion", "mturn_a_nofb", "mturn_m_nofb")] + [
        (k, C.c_int) for k in ("scaling_median", "upper_stellar_turnover", "use_mini_halos", "use_xray")]


class HaloBoxSpec(_Base):
    """``c21cm_halobox_spec`` (include/c21cm_grid.h)."""

    _fields_ = [
        ("dim", C.c_int), ("dim_z", C.c_int), ("hii_dim", C.c_int), ("hii_dim_z", C.c_int),
        ("box_len", C.c_double), ("box_len_z", C.c_double),
        ("perturb_on_high_res", C.c_int), ("lpt2", C.c_int),
        ("growth_factor", C.c_double), ("init_growth_factor", C.c_double),
        ("tab_min", C.c_double), ("tab_width", C.c_double),
        ("ln_nion_table", c_float_p), ("ln_sfrd_table", c_float_p),
        ("prefactor_nion", C.c_double), ("prefactor_sfr", C.c_double),
        ("prefactor_wsfr", C.c_double),
        ("ln_xray_table", c_float_p), ("prefactor_xray", C.c_double),
        # USE_MINI_HALOS
        ("use_mini_halos", C.c_int),
        ("log10_mturn_acg", c_float_p), ("log10_mturn_mcg", c_float_p),
        ("ln_nion_table2d", c_float_p), ("ln_nion_mini_table2d", c_float_p),
        ("mta_min", C.c_double), ("mta_width", C.c_double),
        ("mtm_min", C.c_double), ("mtm_width", C.c_double),
        ("ln_sfrd_mini_table2d", c_float_p), ("ln_xray_table2d", c_float_p),
        ("mt_fixed_min", C.c_double), ("mt_fixed_width", C.c_double),
        ("prefactor_nion_mini", C.c_double), ("prefactor_sfr_mini", C.c_double),
        # halo-catalogue branch
        ("halos", C.POINTER(HaloCatalogStruct)), ("halo_consts", C.POINTER(HaloConsts)),
        ("skip_integral", C.c_int),
    ]


class BrightnessTempStruct(_Base):
    _fields_ = [("brightness_temp", c_float_p), ("tau_21", c_float_p)]


class BrightnessSpec(_Base):
    """``c21cm_brightness_spec`` (include/c21cm_grid.h)."""

    _fields_ = [
        ("n_cells", C.c_size_t),
        ("redshift", C.c_double),
        ("const_factor", C.c_float),
        ("T_rad", C.c_float),
        ("use_ts_fluct", C.c_int),
    ]


def brightness_spec(n_cells, redshift, cosmo=None, use_ts_fluct=False) -> "BrightnessSpec":
    """The two float constants of BrightnessTemperatureBox.c:43-49 for a CosmoParams struct
    (default cosmology if None)."""
    import numpy as np

    cp = cosmo if cosmo is not None else default_cosmo_params()
    z = np.float32(redshift)
    omb, h, omm = np.float32(cp.OMb), np.float32(cp.hlittle), np.float32(cp.OMm)
    const = 27 * (float(omb * h * h) / 0.023) * np.sqrt(
        (0.15 / float(omm) / float(h) / float(h)) * (1.0 + float(z)) / 10.0)
    t_rad = np.float32(2.7255 * float(np.float32(1) + z))
    return BrightnessSpec(n_cells=n_cells, redshift=float(z), const_factor=float(np.float32(const)),
                          T_rad=float(t_rad), use_ts_fluct=int(bool(use_ts_fluct)))


TABLE_FN = C.CFUNCTYPE(C.c_int, C.c_int, C.c_double, C.c_double, c_float_p, C.c_void_p)
# c21cm_table2d_fn: (r_index, prev, dens_min, dens_max, l10mt_min, l10mt_max, l10mt_min_mini,
# l10mt_max_mini, table_acg, table_mcg, user)
TABLE2D_FN = C.CFUNCTYPE(C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double,
                         C.c_double, C.c_double, c_float_p, c_float_p, C.c_void_p)
NDELTA_TABLE, NMTURN_TABLE = 400, 50  # include/c21cm_grid.h


class IonizeSpec(_Base):
    """``c21cm_ionize_spec`` (include/c21cm_grid.h)."""

    _fields_ = [
        ("hii_dim", C.c_int),
        ("hii_dim_z", C.c_int),
        ("box_len", C.c_double),
        ("box_len_z", C.c_double),
        ("n_radii", C.c_int),
        ("r_lowest", C.c_int),
        ("R", C.c_double * MAX_RADII),
        ("sigma_maxmass", C.c_double * MAX_RADII),
        ("hii_filter", C.c_int),
        ("stars_filter", C.c_int),
        ("mfp_meandens", C.c_double),
        ("fcoll_mode", C.c_int),
        ("fix_mean", C.c_int),
        ("mass_dep_zeta", C.c_int),
        ("table_fn", TABLE_FN),
        ("table_user", C.c_void_p),
        ("use_ts_fluct", C.c_int),
        ("recomb_model", C.c_int),
        ("cell_recomb", C.c_int),
        ("minimize_memory", C.c_int),
        ("first_snapshot", C.c_int),
        ("redshift", C.c_double),
        ("stored_redshift", C.c_double),
        ("photoncons_adjustment_factor", C.c_double),
        ("ion_eff_factor", C.c_double),
        ("mean_f_coll", C.c_double),
        ("f_limit_acg", C.c_double),
        ("gamma_prefactor", C.c_double),
        ("rhocrit_omb", C.c_double),
        ("growth_factor", C.c_double),
        ("sigma_minmass", C.c_double),
        ("delta_c", C.c_double),
        ("TK_nofluct", C.c_double),
        ("adia_TK_term", C.c_double),
        ("T_re", C.c_double),
        ("fabs_dtdz", C.c_double),
        ("dz", C.c_double),
        ("rr_y", c_double_p),  # recombination-rate table [RR_NZ][RR_NGAMMA] and its spline c's
        ("rr_c", c_double_p),
        # mini-halos (E-INTEGRAL with USE_MINI_HALOS)
        ("use_mini_halos", C.c_int),
        ("need_prev_ion", C.c_int),
        ("ion_eff_factor_mini", C.c_double),
        ("mean_f_coll_mini", C.c_double),
        ("f_limit_mcg", C.c_double),
        ("gamma_prefactor_mini", C.c_double),
        ("prev_density", c_float_p),
        ("log10_mturn_acg", c_float_p),
        ("log10_mturn_mcg", c_float_p),
        ("table2d_fn", TABLE2D_FN),
        ("table2d_user", C.c_void_p),
        ("ionise_entire_sphere", C.c_int),
    ]


RR_NZ, RR_NGAMMA = 300, 250  # include/c21cm_grid.h C21CM_RR_*


class IonizeReport(_Base):
    _fields_ = [
        ("f_coll_grid_mean", C.c_double * MAX_RADII),
        ("global_xH", C.c_double),
        ("mean_f_coll_out", C.c_double),
        ("ms_preloop", C.c_double),
        ("ms_rloop", C.c_double),
        ("ms_postloop", C.c_double),
        ("f_coll_grid_mean_mini", C.c_double * MAX_RADII),
        ("mean_f_coll_mini_out", C.c_double),
    ]


class MturnSpec(_Base):
    """``c21cm_mturn_spec`` (include/c21cm_grid.h)."""

    _fields_ = [
        ("hii_dim", C.c_int),
        ("hii_dim_z", C.c_int),
        ("first_snapshot", C.c_int),
        ("redshift", C.c_double),
        ("mturn_a_nofb", C.c_double),
        ("mturn_m_nofb", C.c_double),
        ("vcb_const", C.c_double),
        ("A_LW", C.c_double),
        ("BETA_LW", C.c_double),
        ("A_VCB", C.c_double),
        ("BETA_VCB", C.c_double),
        ("sigma_vcb", C.c_double),
    ]


class PerturbSpec(_Base):
    _fields_ = [
        ("dim", C.c_int),
        ("dim_z", C.c_int),
        ("hii_dim", C.c_int),
        ("hii_dim_z", C.c_int),
        ("box_len", C.c_double),
        ("box_len_z", C.c_double),
        ("perturb_algorithm", C.c_int),
        ("perturb_on_high_res", C.c_int),
        ("keep_3d_velocities", C.c_int),
        ("smooth_evolved_density", C.c_int),
        ("density_smooth_radius_mpc", C.c_double),
        ("growth_factor", C.c_double),
        ("init_growth_factor", C.c_double),
        ("dDdt_over_D", C.c_double),
    ]


class IcsSpec(_Base):
    _fields_ = [
        ("dim", C.c_int),
        ("dim_z", C.c_int),
        ("hii_dim", C.c_int),
        ("hii_dim_z", C.c_int),
        ("box_len", C.c_double),
        ("box_len_z", C.c_double),
        ("volume", C.c_float),
        ("perturb_algorithm", C.c_int),
        ("perturb_on_high_res", C.c_int),
        ("density_is_input", C.c_int),
        ("n_m", C.c_int),
        ("pk_by_m", c_double_p),
        ("seed", C.c_ulonglong),
        ("rng_stream", C.c_int),   # 0 Philox (device), 1 the reference's GSL streams
        ("rng_threads", C.c_int),  # N_THREADS the GSL streams are laid out for (1 or 2)
        ("vcb_by_m", c_double_p),  # V_CB_MODEL = FLUCTS: sqrt(P_vcb/P) c / k per |k|^2 index
    ]


# --------------------------------------------------------------------------------------
# Defaults (what lands in the C structs for py21cmfast's default inputs).
# reference: src/py21cmfast/wrapper/inputs.py:492-601 (cosmo), :1014-1111 (simulation),
#            :766-830 (matter), :1302-1355 (astro options), :1569-1680 (astro params).
# --------------------------------------------------------------------------------------
# The reference's Planck18 is astropy's Planck15 with Om0 = (0.02242 + 0.11933) / 0.6766^2 and
# Ob0 = 0.02242 / 0.6766^2 (inputs.py:126-134: Planck 2018 Table 2, last column) -- NOT the rounded
# 0.30966 / 0.04897 of astropy's own Planck18.  The reference's golden power spectra tell the two
# apart: a tilt of 1e-4 over the k range of the 100 Mpc box (tests/test_reference_fixtures.py).
PLANCK18_H = 0.6766
PLANCK18_OMM = (0.02242 + 0.11933) / 0.6766**2
PLANCK18_OMB = 0.02242 / 0.6766**2


def default_cosmo_params(**kw) -> CosmoParams:
    p = CosmoParams(
        hlittle=PLANCK18_H, OMm=PLANCK18_OMM, OMl=1.0 - PLANCK18_OMM, OMb=PLANCK18_OMB,
        POWER_INDEX=0.9665,
        OMn=0.0, OMk=0.0, OMr=8.6e-5, OMtot=1.0, Y_He=0.24, wl=-1.0,
    )
    return p.update(**kw)


def default_simulation_options(**kw) -> SimulationOptions:
    hii_dim = kw.get("HII_DIM", 256)
    p = SimulationOptions(
        HII_DIM=hii_dim, DIM=3 * hii_dim, BOX_LEN=1.5 * hii_dim, NON_CUBIC_FACTOR=1.0,
        N_THREADS=1, Z_HEAT_MAX=35.0, ZPRIME_STEP_FACTOR=1.02, SAMPLER_MIN_MASS=1e8,
        SAMPLER_BUFFER_FACTOR=2.0, N_COND_INTERP=200, N_PROB_INTERP=400, MIN_LOGPROB=-12.0,
        HALOMASS_CORRECTION=0.89, PARKINSON_G0=1.0, PARKINSON_y1=0.0, PARKINSON_y2=0.0,
        INITIAL_REDSHIFT=300.0, DELTA_R_FACTOR=1.1, DENSITY_SMOOTH_RADIUS=0.2,
        DEXM_OPTIMIZE_MINMASS=1e11, DEXM_R_OVERLAP=2.0, CORR_STAR=0.5, CORR_SFR=0.2,
        CORR_LX=0.2, MIN_XE_FOR_FCOLL_IN_TAUX=1e-3,
    )
    return p.update(**kw)


def default_matter_options(**kw) -> MatterOptions:
    p = MatterOptions(
        USE_FFTW_WISDOM=False, HMF=1, V_CB_MODEL=0, POWER_SPECTRUM=0, USE_INTERPOLATION_TABLES=2,
        PERTURB_ON_HIGH_RES=False, PERTURB_ALGORITHM=2, MINIMIZE_MEMORY=False,
        KEEP_3D_VELOCITIES=False, DEXM_OPTIMIZE=False, FILTER=0, HALO_FILTER=0,
        SMOOTH_EVOLVED_DENSITY_FIELD=False, SOURCE_MODEL=4, SAMPLE_METHOD=0,
    )
    return p.update(**kw)


def default_astro_params(**kw) -> AstroParams:
    p = AstroParams(
        HII_EFF_FACTOR=30.0, F_STAR10=10 ** -1.3, ALPHA_STAR=0.5, ALPHA_STAR_MINI=0.0,
        SIGMA_STAR=0.25, UPPER_STELLAR_TURNOVER_MASS=10 ** 11.447,
        UPPER_STELLAR_TURNOVER_INDEX=-0.6, F_STAR7_MINI=10 ** -2.0, t_STAR=0.5,
        SIGMA_SFR_INDEX=-0.12, SIGMA_SFR_LIM=0.19, L_X=10 ** 40.5, L_X_MINI=10 ** 40.5,
        SIGMA_LX=0.5, F_ESC10=10 ** -1.0, ALPHA_ESC=-0.5, F_ESC7_MINI=10 ** -2.0, T_RE=2e4,
        M_TURN=10 ** 8.7, R_BUBBLE_MAX=15.0, ION_Tvir_MIN=10 ** 4.69897, F_H2_SHIELD=0.0,
        NU_X_THRESH=500.0, X_RAY_SPEC_INDEX=1.0, X_RAY_Tvir_MIN=10 ** 4.69897, A_LW=2.0,
        BETA_LW=0.6, A_VCB=1.0, BETA_VCB=1.8, V_CB_AVG_DEBUG=25.86, POP2_ION=5000.0,
        POP3_ION=44021.0, PHOTONCONS_CALIBRATION_END=3.5, CLUMPING_FACTOR=2.0, ALPHA_UVB=5.0,
        R_MAX_TS=500.0, N_STEP_TS=40, DELTA_R_HII_FACTOR=1.1, R_BUBBLE_MIN=0.620350491,
        MAX_DVDR=0.2, NU_X_MAX=10000.0, NU_X_BAND_MAX=2000.0,
    )
    return p.update(**kw)


def default_astro_options(**kw) -> AstroOptions:
    p = AstroOptions(
        USE_MINI_HALOS=False, USE_X_RAY_HEATING=True, USE_CMB_HEATING=True, USE_LYA_HEATING=True,
        RECOMB_MODEL=0, USE_TS_FLUCT=False, M_MIN_in_Mass=True, USE_EXP_FILTER=True,
        CELL_RECOMB=True, LYA_MULTIPLE_SCATTERING=False, USE_ADIABATIC_FLUCTUATIONS=True,
        PHOTON_CONS_TYPE=0, USE_UPPER_STELLAR_TURNOVER=True, HALO_SCALING_RELATIONS_MEDIAN=False,
        HII_FILTER=0, HEAT_FILTER=0, IONISE_ENTIRE_SPHERE=False, INTEGRATION_METHOD_ATOMIC=1,
        INTEGRATION_METHOD_MINI=1,
    )
    return p.update(**kw)


def default_cosmo_tables(**kw) -> CosmoTables:
    p = CosmoTables(ps_norm=0.8102, USE_SIGMA_8=True, V_CB_AVG=25.86)
    return p.update(**kw)


MAX_TS_RADII = 128
MAX_ANNULAR_GRIDS = 5


class RboxSpec(_Base):
    """``c21cm_rbox_spec`` (include/c21cm_grid.h)."""

    _fields_ = [
        ("hii_dim", C.c_int),
        ("hii_dim_z", C.c_int),
        ("box_len", C.c_double),
        ("box_len_z", C.c_double),
        ("filter_type", C.c_int),
        ("n_R", C.c_int),
        ("R", C.c_double * MAX_TS_RADII),
        ("cell_radius", C.c_double),
        ("min_value", C.c_double),
        ("const_factor", C.c_double),
    ]


def rbox_spec(hii_dim, box_len, radii, filter_type=0, min_value=-1.0, const_factor=1.0,
              hii_dim_z=None, box_len_z=None) -> "RboxSpec":
    """fill_Rbox_table's arguments; cell_radius = L_FACTOR BOX_LEN / HII_DIM
    (SpinTemperatureBox.c:585-588, Constants.c:41)."""
    s = RboxSpec(hii_dim=hii_dim, hii_dim_z=hii_dim_z or hii_dim, box_len=box_len,
                 box_len_z=box_len_z or box_len, filter_type=filter_type, n_R=len(radii),
                 cell_radius=0.620350491 * (box_len / hii_dim), min_value=min_value,
                 const_factor=const_factor)
    for i, r in enumerate(radii):
        s.R[i] = float(r)
    return s


X_INT_NXHII = 14
X_INT_XHII = (1.0e-4, 2.318e-4, 4.677e-4, 1.0e-3, 2.318e-3, 4.677e-3, 1.0e-2, 2.318e-2, 4.677e-2,
              1.0e-1, 0.5, 0.9, 0.99, 0.999)  # elec_interp.c:57-70 (floats upstream)
LYA_NT, LYA_NGP = 101, 51
TS_SRC_GRIDS, TS_SRC_SFRD_TABLE, TS_SRC_FCOLL_TABLES = 0, 1, 2
_PER_SHELL = C.c_double * MAX_TS_RADII


class TsSpec(_Base):
    """``c21cm_ts_spec`` (include/c21cm_grid.h): scalars and tables of the per-cell part of
    ComputeTsBox."""

    _fields_ = [
        ("hii_dim", C.c_int), ("hii_dim_z", C.c_int), ("n_step", C.c_int),
        ("source_mode", C.c_int),
        ("use_xray_heating", C.c_int), ("use_cmb_heating", C.c_int), ("use_lya_heating", C.c_int),
        ("no_light", C.c_int),
        ("redshift", C.c_double), ("dzp", C.c_double), ("growth_ratio", C.c_double),
        ("No", C.c_double), ("N_b0", C.c_double), ("h_frac", C.c_double), ("he_frac", C.c_double),
        ("k_B", C.c_double), ("h_p", C.c_double), ("m_p", C.c_double), ("c_cms", C.c_double),
        ("A10", C.c_double), ("T_21", C.c_double), ("lambda_21", C.c_double),
        ("nu_Ly_alpha", C.c_double),
        ("clumping_factor", C.c_double),
        ("xray_prefactor", C.c_double), ("Trad", C.c_double), ("Ts_prefactor", C.c_double),
        ("xa_tilde_prefactor", C.c_double), ("xc_inverse", C.c_double),
        ("dcomp_dzp_prefactor", C.c_double),
        ("Nb_zp", C.c_double), ("N_zp", C.c_double), ("lya_star_prefactor", C.c_double),
        ("volunit_inv", C.c_double), ("hubble_zp", C.c_double), ("growth_zp", C.c_double),
        ("dgrowth_dzp", C.c_double), ("dt_dzp", C.c_double),
        ("z_edge_factor", _PER_SHELL), ("xray_R_factor", _PER_SHELL),
        ("starlya_prefactor", _PER_SHELL), ("lya_cont_prefactor", _PER_SHELL),
        ("lya_inj_prefactor", _PER_SHELL),
        ("zpp_growth", _PER_SHELL), ("mean_sfr_zpp", _PER_SHELL),
        ("tab_min", _PER_SHELL), ("tab_width", _PER_SHELL),
        ("ln_sfrd_tables", c_float_p), ("fcoll_tables", c_float_p), ("dfcoll_tables", c_float_p),
        ("sfr_scale", C.c_double), ("xray_scale", C.c_double),
        ("freq_int_heat", c_double_p), ("freq_int_ion", c_double_p), ("freq_int_lya", c_double_p),
        ("lya_dEC", c_double_p), ("lya_dEI", c_double_p),
        # USE_MINI_HALOS (SFRD_TABLE mode)
        ("use_mini_halos", C.c_int),
        ("starlya_prefactor_mini", _PER_SHELL), ("lya_cont_prefactor_mini", _PER_SHELL),
        ("lya_inj_prefactor_mini", _PER_SHELL), ("lw_prefactor", _PER_SHELL),
        ("lw_prefactor_mini", _PER_SHELL), ("mean_sfr_zpp_mini", _PER_SHELL),
        ("ln_sfrd_tables_mini", c_float_p),
        ("mturn_tab_min", C.c_double), ("mturn_tab_width", C.c_double),
        ("sfr_scale_mini", C.c_double), ("xray_scale_mini", C.c_double),
        ("filtered_log10_mcrit", c_float_p),
    ]


class TsReport(_Base):
    """``c21cm_ts_report``."""

    _fields_ = [("Ts_ave", C.c_double), ("Tk_ave", C.c_double), ("x_e_ave", C.c_double),
                ("J_alpha_ave", C.c_double), ("xheat_ave", C.c_double), ("xion_ave", C.c_double),
                ("ave_sfrd", _PER_SHELL), ("ave_sfrd_mini", _PER_SHELL)]


class TsFirstSpec(_Base):
    """``c21cm_ts_first_spec`` (init_first_Ts)."""

    _fields_ = [("hii_dim", C.c_int), ("hii_dim_z", C.c_int), ("redshift", C.c_double),
                ("perturbed_redshift", C.c_double), ("inverse_growth_factor_z", C.c_float),
                ("growth_factor_zp", C.c_float), ("xe", C.c_double), ("TK", C.c_double),
                ("cT_ad", C.c_double), ("No", C.c_double), ("N_b0", C.c_double),
                ("A10", C.c_double), ("T_21", C.c_double), ("T_cmb", C.c_double)]


class AnnularSpec(_Base):
    """``c21cm_annular_spec`` (include/c21cm_grid.h)."""

    _fields_ = [
        ("hii_dim", C.c_int),
        ("hii_dim_z", C.c_int),
        ("box_len", C.c_double),
        ("box_len_z", C.c_double),
        ("R_inner", C.c_double),
        ("R_outer", C.c_double),
        ("R_star", C.c_double),
        ("n_grids", C.c_int),
        ("filter_type", C.c_int * MAX_ANNULAR_GRIDS),
    ]


def annular_spec(hii_dim, box_len, R_inner, R_outer, filter_types, R_star=0.0, hii_dim_z=None,
                 box_len_z=None) -> "AnnularSpec":
    s = AnnularSpec(hii_dim=hii_dim, hii_dim_z=hii_dim_z or hii_dim, box_len=box_len,
                    box_len_z=box_len_z or box_len, R_inner=R_inner, R_outer=R_outer,
                    R_star=R_star, n_grids=len(filter_types))
    for i, t in enumerate(filter_types):
        s.filter_type[i] = int(t)
    return s


class XraySourceBoxStruct(_Base):
    """``XraySourceBox`` (include/c21cm_abi.h; reference _outputstructs_wrapper.h:67-77)."""

    _fields_ = [
        ("filtered_sfr", c_float_p),
        ("filtered_xray", c_float_p),
        ("filtered_sfr_mini", c_float_p),
        ("filtered_sfr_lw", c_float_p),
        ("filtered_sfr_mini_lw", c_float_p),
        ("mean_log10_Mcrit_LW", C.POINTER(C.c_double)),
        ("mean_sfr", C.POINTER(C.c_double)),
        ("mean_sfr_mini", C.POINTER(C.c_double)),
    ]


def class_tables(k, transfer_density, transfer_vcb=None, **kw) -> CosmoTables:
    """CosmoTables holding CLASS-style transfer functions (POWER_SPECTRUM = CLASS): T(k) =
    delta(k, z=0) / zeta(k) on an ascending k grid [1/Mpc], optionally the DM-baryon relative
    velocity transfer on the same grid.  The numpy arrays are kept alive on the returned object."""
    import numpy as np

    k = np.ascontiguousarray(k, np.float64)
    td = np.ascontiguousarray(transfer_density, np.float64)
    t1 = Table1D(size=len(k), x_values=k.ctypes.data_as(c_double_p),
                 y_values=td.ctypes.data_as(c_double_p))
    ct = default_cosmo_tables(**kw)
    ct.transfer_density = C.pointer(t1)
    keep = [k, td, t1]
    if transfer_vcb is not None:
        tv = np.ascontiguousarray(transfer_vcb, np.float64)
        t2 = Table1D(size=len(k), x_values=k.ctypes.data_as(c_double_p),
                     y_values=tv.ctypes.data_as(c_double_p))
        ct.transfer_vcb = C.pointer(t2)
        keep += [tv, t2]
    ct._keep = keep
    return ct
