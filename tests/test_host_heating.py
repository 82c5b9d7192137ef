"""Host scalars of the spin-temperature calculation (pure C host code of lib21cmfast_hip.so, no
GPU needed) against independent numpy / scipy evaluations (oracle/ref_heating.py) with the
reference's own data tables (tests/golden/reference/_data).

reference: src/py21cmfast/src/heating_helper_progs.c (spectral_emissivity, the frequency
integrals, tauX, nu_tau_one), elec_interp.c, SpinTemperatureBox.c:312-499,810-1008,1098-1184.

The reference integrates with gsl_integration_qag at loose tolerances (1e-2, 5e-3) and brackets
the tau_X = 1 frequency to 2 %; the library restates QAG and Brent so that it stops where the
reference stops.  Here those routines are checked on closed forms at tight tolerances, and the
physics tables against converged scipy quadrature within the tolerance the reference asks for."""

import ctypes as C
import math
from pathlib import Path

import numpy as np
import pytest
from scipy import integrate

from oracle import ref_heating as RH
from oracle.ref_scalars import Cosmo

DATA = Path(__file__).parent / "golden" / "reference" / "_data"
f64, f32, i32 = C.c_double, C.c_float, C.c_int
FN = C.CFUNCTYPE(f64, f64, C.c_void_p)


@pytest.fixture(scope="module")
def heat(pkg):
    lib = pkg.load()
    S = pkg.structs
    keep = dict(so=S.default_simulation_options(HII_DIM=50, DIM=150, BOX_LEN=100.0, Z_HEAT_MAX=35.0),
                mo=S.default_matter_options(SOURCE_MODEL=1), cp=S.default_cosmo_params(),
                ap=S.default_astro_params(), ao=S.default_astro_options(USE_LYA_HEATING=False),
                ct=S.default_cosmo_tables())
    lib.Broadcast_struct_global_all(*(C.byref(keep[k]) for k in ("so", "mo", "cp", "ap", "ao", "ct")))
    keep["path"] = str(DATA).encode()
    S.ConfigSettings.in_dll(lib, "config_settings").external_table_path = keep["path"]
    lib.init_ps.restype = None
    lib.init_ps()
    sig = {
        "c21_qag15": (f64, [FN, C.c_void_p, f64, f64, f64, C.POINTER(f64), C.POINTER(i32)]),
        "c21_brent_root": (f64, [FN, C.c_void_p, f64, f64, f64, i32, C.POINTER(i32)]),
        "c21_frecycle": (f64, [i32]), "c21_nu_n": (f64, [i32]), "c21_zmax": (f32, [f32, i32]),
        "c21_spectral_emissivity": (f64, [f64, i32]),
        "c21_HI_ion_crosssec": (f64, [f64]), "c21_HeI_ion_crosssec": (f64, [f64]),
        "c21_HeII_ion_crosssec": (f64, [f64]),
        "c21_nu_integrand": (f64, [f64, f64, i32]),
        "c21_integrate_over_nu": (f64, [f64, f64, f64, i32]),
        "c21_EvaluateNionTs": (f64, [f64]), "c21_EvaluateSFRD": (f64, [f64]),
        "c21_tauX": (f64, [f64] * 6), "c21_nu_tau_one": (f64, [f64, f64, f64, f64, C.POINTER(i32)]),
        "c21_minimum_source_mass_xray": (f64, [f64]),
        "c21_heat_load": (i32, []), "init_heat": (i32, []),
        "c21_ts_prepare": (i32, [f32, f32, f32, f64, C.c_void_p, C.c_void_p]),
    }
    for k in ("fheat", "n_Lya", "nion_HI", "nion_HeI", "nion_HeII"):
        sig[f"c21_interp_{k}"] = (f32, [f32, f32])
    for name, (res, args) in sig.items():
        getattr(lib, name).restype = res
        getattr(lib, name).argtypes = args
    assert lib.c21_heat_load() == 0, pkg.last_error()
    lib._keep = keep
    return lib


# ------------------------------------------------------------------ the two GSL restatements
def test_qag15_against_closed_forms(heat):
    cases = [  # (f, a, b, exact)
        (lambda x: x * x, 0.0, 1.0, 1 / 3),
        (lambda x: math.sqrt(x), 0.0, 1.0, 2 / 3),                  # endpoint singular derivative
        (lambda x: math.exp(-x * x), -3.0, 5.0, math.sqrt(math.pi) / 2 * (math.erf(3) + math.erf(5))),
        (lambda x: math.cos(40 * x) / (1 + x), 0.0, 2.0, None),    # oscillatory: many bisections
        (lambda x: abs(x - 0.3) ** 0.5, 0.0, 1.0, (0.3**1.5 + 0.7**1.5) / 1.5),  # interior kink
    ]
    for fpy, a, b, exact in cases:
        cb = FN(lambda x, _: fpy(x))
        for eps in (1e-2, 1e-6, 1e-10):
            err, st = f64(), i32()
            got = heat.c21_qag15(cb, None, a, b, eps, C.byref(err), C.byref(st))
            want = exact if exact is not None else integrate.quad(fpy, a, b, limit=500, epsabs=0,
                                                                 epsrel=1e-13)[0]
            assert st.value == 0
            # QUADPACK's contract: the error estimate honours the request and bounds the error
            assert err.value <= eps * abs(got) * (1 + 1e-12)
            assert abs(got - want) <= max(err.value, 1e-14 * abs(want))
    # a polynomial of degree <= 22 is integrated exactly by the first 15-point rule
    cb = FN(lambda x, _: 3 * x**10 - x**3)
    assert heat.c21_qag15(cb, None, -1.0, 2.0, 1e-2, None, None) == pytest.approx(
        3 * (2**11 + 1) / 11 - (16 - 1) / 4, rel=1e-14)


def test_brent_root_brackets_like_gsl(heat):
    for fpy, lo, hi, root in ((lambda x: x * x - 2, 0.0, 3.0, math.sqrt(2)),
                              (lambda x: math.exp(x) - 5, -2.0, 9.0, math.log(5)),
                              (lambda x: 1e-3 * (x - 1e15) ** 3, 1e12, 1e18, 1e15)):
        cb = FN(lambda x, _: fpy(x))
        st = i32()
        r = heat.c21_brent_root(cb, None, lo, hi, 1e-12, 200, C.byref(st))
        assert st.value == 0 and r == pytest.approx(root, rel=1e-10)
        # a loose interval test stops early: the iterate is inside a 2 % bracket of the root
        r2 = heat.c21_brent_root(cb, None, lo, hi, 0.02, 100, C.byref(st))
        assert st.value == 0 and abs(r2 / root - 1) < 0.03
    st = i32()
    heat.c21_brent_root(FN(lambda x, _: x * x + 1), None, 0.0, 1.0, 1e-6, 50, C.byref(st))
    assert st.value == 1  # endpoints do not straddle zero


# ------------------------------------------------------------------ spectra, tables, cross sections
def test_stellar_spectra_and_recycling(heat):
    sp = RH.StellarSpectra(DATA / "stellar_spectra.dat")
    for n in range(0, 33):
        assert heat.c21_frecycle(n) == RH.frecycle(n)
    for n in range(2, 24):
        assert heat.c21_nu_n(n) == pytest.approx(RH.nu_n(n), rel=1e-15)
        assert heat.c21_zmax(12.0, n) == pytest.approx(RH.zmax(12.0, n), rel=1e-6)
    for nu in np.linspace(0.76, 1.33, 60):
        assert heat.c21_spectral_emissivity(nu, 2) == pytest.approx(sp.emissivity(nu, 2), rel=2e-6)
        assert heat.c21_spectral_emissivity(nu, 3) == pytest.approx(sp.emissivity(nu, 3), rel=2e-6)
    # normalisation (:342-347): band i integrates to its tabulated photon number times POP2_ION
    raw = np.array([ln.split() for ln in open(DATA / "stellar_spectra.dat").read().splitlines()[:22]],
                   float)
    for i in (1, 2, 7):  # (the narrow high-n bands feel the float rounding of their edges)
        band = integrate.quad(lambda x: heat.c21_spectral_emissivity(x, 2) * RH.PC["nu_Ly_alpha"],
                              float(sp.nu[i]) * (1 + 1e-7), float(sp.nu[i + 1]) * (1 - 1e-7))[0]
        assert band == pytest.approx(raw[i - 1, 1] * 5000.0, rel=1e-4)


def test_x_int_lookups_and_cross_sections(heat):
    t = RH.XIntTables(DATA / "x_int_tables")
    rng = np.random.default_rng(3)
    for _ in range(200):
        En = float(10 ** rng.uniform(0.8, 4.1))
        x = float(10 ** rng.uniform(-4.3, 0))
        for k in RH.XIntTables.FIELDS:
            got = getattr(heat, f"c21_interp_{k}")(En, x)
            assert got == pytest.approx(t.interp(k, En, x), rel=3e-6, abs=1e-12), (k, En, x)
    for nu in (2e15, 3.3e15, 6e15, 1.4e16, 1e17, 1e18):
        assert heat.c21_HI_ion_crosssec(nu) == pytest.approx(RH.HI_ion_crosssec(nu), rel=1e-12)
        assert heat.c21_HeI_ion_crosssec(nu) == pytest.approx(RH.HeI_ion_crosssec(nu), rel=1e-12)
        assert heat.c21_HeII_ion_crosssec(nu) == pytest.approx(RH.HeII_ion_crosssec(nu), rel=1e-12)
    assert heat.c21_HI_ion_crosssec(RH.PC["nu_ion_HI"] * (1 + 1e-9)) == pytest.approx(6.3e-18, rel=1e-6)


def test_frequency_integrals(heat):
    c, t = Cosmo(), RH.XIntTables(DATA / "x_int_tables")
    d = RH.densities(c)
    lo = 500.0 * RH.PC["eV_to_Hz"]
    for nu in (1.3e17, 4e17, 2e18):
        for flag in (0, 1, 2):
            assert heat.c21_nu_integrand(nu, 2.318e-3, flag) == pytest.approx(
                RH.nu_integrand(t, d, nu, 2.318e-3, flag), rel=5e-6)
    for x_e in (1e-4, 1e-2, 0.5):
        for flag in (0, 1, 2):
            for lower in (lo, 2.2 * lo):
                got = heat.c21_integrate_over_nu(12.0, x_e, lower, flag)
                want = RH.integrate_over_nu(t, d, c, 12.0, x_e, lower, flag)
                assert got == pytest.approx(want, rel=1e-2), (x_e, flag)  # the reference's epsrel


# ------------------------------------------------------------------ the per-snapshot preparation
class Tables(C.Structure):
    _R = f64 * 128
    _fields_ = [("n_step", i32), ("no_light", i32), ("Q_HI", f64)] + [
        (k, f64 * 128) for k in ("R_values", "zpp_edge", "zpp", "dzpp", "dtdz", "zpp_growth",
                                 "M_min_R", "M_max_R", "starlya_prefactor", "lya_cont_prefactor",
                                 "lya_inj_prefactor", "mean_sfr_zpp", "nu_tau_one")
    ] + [("freq", C.POINTER(f64)), ("sfrd_tables", C.POINTER(f32)), ("fcoll_tables", C.POINTER(f32)),
         ("dfcoll_tables", C.POINTER(f32)), ("sigma_min", f64 * 128), ("sigma_max", f64 * 128)] + [
        (k, f64 * 128) for k in ("ave_log10_mturn", "mean_sfr_zpp_mini", "starlya_prefactor_mini",
                                 "lya_cont_prefactor_mini", "lya_inj_prefactor_mini", "lw_prefactor",
                                 "lw_prefactor_mini")
    ] + [("sfrd_tables_mini", C.POINTER(f32)), ("shell_mask", C.c_void_p)]


def test_ts_prepare_against_numpy(heat, pkg):
    S = pkg.structs
    zp, prev_z, x_e_ave = 18.0, 18.76, 2.3e-4
    spec, tab = S.TsSpec(), Tables()
    assert heat.c21_ts_prepare(zp, prev_z, zp, x_e_ave, C.byref(spec), C.byref(tab)) == 0, pkg.last_error()
    n = tab.n_step
    assert n == 40 and spec.n_step == 40 and spec.source_mode == S.TS_SRC_SFRD_TABLE
    c = Cosmo()
    d = RH.densities(c)
    # shells
    ze = RH.z_edges(c, zp, 50, 100.0, n_step=40, R_MAX_TS=500.0)
    for k, name in (("R", "R_values"), ("zpp_edge", "zpp_edge"), ("zpp", "zpp"), ("dzpp", "dzpp"),
                    ("dtdz", "dtdz"), ("growth", "zpp_growth")):
        np.testing.assert_allclose(np.array(getattr(tab, name)[:n]), ze[k], rtol=2e-6, err_msg=k)
    assert tab.R_values[0] == pytest.approx(0.620350491 * 2.0) and tab.R_values[n - 1] < 500.0
    # stellar Lyman-alpha prefactors
    sp = RH.StellarSpectra(DATA / "stellar_spectra.dat")
    sf = RH.spectral_factors(sp, zp, np.array(tab.zpp[:n]))
    for k, name in (("starlya", "starlya_prefactor"), ("cont", "lya_cont_prefactor"),
                    ("inj", "lya_inj_prefactor")):
        np.testing.assert_allclose(np.array(getattr(spec, name)[:n]), sf[k], rtol=3e-6, atol=1e-30,
                                   err_msg=k)
    assert sf["starlya"][0] > 0 and sf["starlya"][-1] == 0  # the outer shells are beyond Ly-beta's horizon
    # z' constants
    zc = RH.zp_consts(c, zp, lagrangian=False)
    for k, v in zc.items():
        tol = 2e-4 if k == "dgrowth_dzp" else 3e-6  # a forward difference over dz = 1e-10
        assert getattr(spec, k) == pytest.approx(v, rel=tol), k
    assert spec.dzp == pytest.approx(zp - prev_z, rel=1e-6) and spec.redshift == zp
    np.testing.assert_allclose(np.array(spec.z_edge_factor[:n]),
                               np.abs(ze["dzpp"] * ze["dtdz"]) * np.array([c.hubble(z) for z in ze["zpp"]]) / 0.5,
                               rtol=3e-6)
    np.testing.assert_allclose(np.array(spec.xray_R_factor[:n]), (1 + ze["zpp"]) ** -1.0, rtol=1e-6)
    # global collapsed fractions: N_ion(z) table against the exported integrals of cosmology.c
    lnMmin = math.log(heat.c21_minimum_source_mass_xray(zp))
    assert math.exp(lnMmin) == pytest.approx(10**8.7 / 50.0, rel=1e-6)
    nion = heat.c21_EvaluateNionTs(zp)
    assert 0 < nion < 1e-2
    ion_eff = 10**-1.3 * 10**-1.0 * 5000.0
    assert tab.Q_HI == pytest.approx(1 - ion_eff * nion / (1 - x_e_ave), rel=1e-9)
    assert tab.no_light == 0 and spec.no_light == 0
    sfrd = np.array(spec.mean_sfr_zpp[:n])
    # both are in units of their normalisations f_*10 (f_esc10): the escape fraction of small
    # haloes exceeds f_esc10 by up to 1 / f_esc10 = 10
    assert np.all(np.diff(sfrd) < 0) and 1 < nion / heat.c21_EvaluateSFRD(zp) < 10
    # tau_X = 1 frequencies and the frequency-integral tables of three shells
    t = RH.XIntTables(DATA / "x_int_tables")
    nion_of_z = heat.c21_EvaluateNionTs
    fn = 14 * n
    freq = np.ctypeslib.as_array(tab.freq, (3 * fn,)).reshape(3, 14, n)
    for R_ct in (0, 17, 39):
        zpp = tab.zpp[R_ct]
        want_nu = RH.nu_tau_one(c, d, zp, zpp, x_e_ave, ion_eff, nion_of_z)
        assert tab.nu_tau_one[R_ct] == pytest.approx(want_nu, rel=0.03), R_ct  # a 2 % bracket
        # tau_X at the library's frequency is 1 within what a 2 % frequency error allows (~ nu^-3)
        if want_nu > RH.PC["nu_ion_HeI"]:
            assert heat.c21_tauX(tab.nu_tau_one[R_ct], x_e_ave, x_e_ave, zp, zpp, ion_eff) == pytest.approx(1.0, rel=0.1)
        lower = max(tab.nu_tau_one[R_ct], 500.0 * RH.PC["eV_to_Hz"])
        for x_ct in (0, 6, 13):
            for flag in (0, 1, 2):
                want = RH.integrate_over_nu(t, d, c, zp, float(RH.X_INT_XHII[x_ct]), lower, flag)
                assert freq[flag, x_ct, R_ct] == pytest.approx(want, rel=1e-2), (R_ct, x_ct, flag)
    assert np.all(freq > 0)
    assert np.all(freq[0, :, 0] > freq[0, :, 39])  # harder spectrum from farther away
    heat.c21_ts_tables_free(C.byref(tab))


def test_unsupported_options_and_missing_tables(heat, pkg, tmp_path):
    S = pkg.structs
    spec, tab = S.TsSpec(), Tables()
    keep = heat._keep
    keep["ao"].USE_MINI_HALOS = True  # built for E-INTEGRAL
    keep["mo"].SOURCE_MODEL = 0
    assert heat.c21_ts_prepare(18.0, 18.7, 18.0, 2e-4, C.byref(spec), C.byref(tab)) == 3
    assert "USE_MINI_HALOS" in pkg.last_error()
    keep["ao"].USE_MINI_HALOS = False
    keep["mo"].SOURCE_MODEL = 1
    keep["ao"].USE_LYA_HEATING = True  # the table is not part of the reference checkout
    assert heat.c21_ts_prepare(18.0, 18.7, 18.0, 2e-4, C.byref(spec), C.byref(tab)) == 1
    assert "Lyman_alpha_heating_table" in pkg.last_error()
    keep["ao"].USE_LYA_HEATING = False
    cfg = S.ConfigSettings.in_dll(heat, "config_settings")
    empty = str(tmp_path).encode()
    cfg.external_table_path = empty
    assert heat.init_heat() != 0 and "recfast" in pkg.last_error().lower()
    cfg.external_table_path = keep["path"]
    assert heat.init_heat() == 0
