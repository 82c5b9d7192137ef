"""Host-side scalars and tables of the molecularly cooled ("mini-halo") population, against scipy
and closed forms.  Reference behaviour: scaling_relations.c:36-119, thermochem.c:281-311,
hmf.c:470-477,973-990,1066-1104, interp_tables.c:291-405 (USE_MINI_HALOS)."""
import ctypes as C
import math

import numpy as np
import pytest
from scipy import integrate

from test_host_scalars import ScalingConsts, _bind_conditional, host  # noqa: F401  (fixture)

f64 = C.c_double


def _bind(lib):
    _bind_conditional(lib)
    lib.c21_lyman_werner_threshold.restype = f64
    lib.c21_lyman_werner_threshold.argtypes = [C.c_float] * 3
    lib.c21_reionization_feedback.restype = f64
    lib.c21_reionization_feedback.argtypes = [C.c_float] * 3
    lib.c21_Nion_General_MINI.restype = f64
    lib.c21_Nion_General_MINI.argtypes = [f64] * 4 + [C.POINTER(ScalingConsts)]
    lib.c21_Nion_General.restype = f64
    lib.c21_Nion_General.argtypes = [f64] * 4 + [C.POINTER(ScalingConsts)]
    lib.c21_Nion_ConditionalM_MINI.restype = f64
    lib.c21_Nion_ConditionalM_MINI.argtypes = [f64] * 7 + [C.POINTER(ScalingConsts), C.c_int]
    lib.c21_Nion_Conditional_table2d.restype = C.c_int
    lib.c21_Nion_Conditional_table2d.argtypes = [f64] * 9 + [C.POINTER(ScalingConsts), C.c_int,
                                                             C.c_int, f64, C.c_int,
                                                             C.POINTER(C.c_float), C.c_int, C.c_int]


@pytest.fixture()
def mini(host, pkg):
    """the host library with USE_MINI_HALOS broadcast (restored afterwards)"""
    S = pkg.structs
    keep = host._keep
    _bind(host)

    def broadcast(ap, ao):
        keep["mini_structs"] = (ap, ao)
        host.Broadcast_struct_global_all(C.byref(keep["so"]), C.byref(keep["mo"]),
                                         C.byref(keep["cp"]), C.byref(ap), C.byref(ao),
                                         C.byref(keep["ct"]))
    ap = S.default_astro_params(ALPHA_STAR_MINI=0.5, F_STAR7_MINI=10 ** -2.5)
    broadcast(ap, S.default_astro_options(USE_MINI_HALOS=True))
    host.mini_ap = ap
    yield host
    broadcast(keep["ap"], keep["ao"])


def test_thresholds(mini):
    ap = mini.mini_ap
    ct = mini._keep["ct"]
    z = 17.0
    # no LW background, no streaming velocity: the molecular-cooling mass of Visbal+15
    assert mini.c21_lyman_werner_threshold(z, 0.0, 0.0) == pytest.approx(3.314e7 * 18.0**-1.5,
                                                                        rel=1e-6)
    j, v = 0.3, 20.0
    sig = ct.V_CB_AVG * math.sqrt(3 * math.pi / 8)
    want = (3.314e7 * 18.0**-1.5 * (1 + ap.A_LW * np.float32(j) ** ap.BETA_LW)
            * (1 + ap.A_VCB * v / sig) ** ap.BETA_VCB)
    assert mini.c21_lyman_werner_threshold(z, j, v) == pytest.approx(want, rel=1e-6)
    # Sobacchi & Mesinger 2013: no feedback in cells that were never ionised
    assert mini.c21_reionization_feedback(8.0, 0.5, -1.0) == 1e-40
    want = 3e9 * (2 * 0.5) ** 0.17 * (9.0 / 10) ** -2.1 * (1 - (9.0 / 11.0) ** 2) ** 2.5
    assert mini.c21_reionization_feedback(8.0, 0.5, 10.0) == pytest.approx(want, rel=1e-6)


def test_scaling_constants(mini):
    ap = mini.mini_ap
    sc = ScalingConsts()
    assert mini.c21_set_scaling_constants(15.0, C.byref(sc)) == 0
    assert sc.alpha_star_mini == pytest.approx(ap.ALPHA_STAR_MINI)
    assert sc.mturn_a_nofb == pytest.approx(max(sc.acg_thresh, ap.M_TURN))
    assert sc.mturn_m_nofb == pytest.approx(3.314e7 * 16.0**-1.5, rel=1e-6)  # V_CB_MODEL none
    # f_*(M) = F_STAR7 (M/1e7)^a reaches one at Mlim (float bisection, 1e-3 in log10 M)
    assert sc.fstar_7 * (sc.Mlim_Fstar_mini / 1e7) ** sc.alpha_star_mini == pytest.approx(1.0, rel=5e-3)
    # a falling f_esc(M) that is below one already at 1e5 Msun is never capped
    assert sc.fesc_7 * (1e5 / 1e7) ** sc.alpha_esc < 1 and sc.Mlim_Fesc_mini == 1e5


def _weight_mini(sc, lnM, Mturn):
    def pl(norm, alpha, lim):
        if (alpha > 0 and lnM > math.log(lim)) or (alpha < 0 and lnM < math.log(lim)):
            return -math.log(norm)
        return alpha * (lnM - 7 * math.log(10))
    M = math.exp(lnM)
    return math.exp(pl(sc.fstar_7, sc.alpha_star_mini, sc.Mlim_Fstar_mini)
                    + pl(sc.fesc_7, sc.alpha_esc, sc.Mlim_Fesc_mini)
                    - M / sc.acg_thresh - Mturn / M + lnM)


def test_nion_general_mini_against_scipy(mini):
    """Sheth-Tormen mass function x the MCG n_ion(M), integrated by scipy with the library's sigma"""
    sc = ScalingConsts()
    z = 14.0
    assert mini.c21_set_scaling_constants(z, C.byref(sc)) == 0
    D = mini.dicke(z)
    Mturn = 3e6

    def mf(lnM):
        M = math.exp(lnM)
        sig = mini.c21_sigma_fast(M) * D
        ds = mini.dsigmasqdm_z0(M) * D * D / (2 * sig)
        nu = math.sqrt(0.73) * 1.686 / sig
        return (-(ds / sig) * math.sqrt(2 / math.pi) * 0.353 * (1 + nu ** (-2 * 0.175)) * nu
                * math.exp(-nu * nu / 2))

    lo, hi = math.log(1e5), math.log(1e16)
    want, _ = integrate.quad(lambda x: _weight_mini(sc, x, Mturn) * mf(x), lo, math.log(1e11),
                             limit=400, epsrel=1e-9)
    got = mini.c21_Nion_General_MINI(z, lo, hi, Mturn, C.byref(sc))
    assert got == pytest.approx(want, rel=2e-4)
    # a stronger LW background (higher turnover) suppresses the population
    assert mini.c21_Nion_General_MINI(z, lo, hi, 3e7, C.byref(sc)) < got


def test_conditional_tables_2d(mini):
    sc = ScalingConsts()
    z = 12.0
    assert mini.c21_set_scaling_constants(z, C.byref(sc)) == 0
    D = mini.dicke(z)
    Mmin, Mcond = 1e5, mini.c21_RtoM(3.0)
    s_c = mini.c21_sigma_fast(Mcond)
    lnMmin, lnMc = math.log(Mmin), math.log(Mcond)
    nd, nm = 400, 50
    dmin, dmax = -0.9, 1.5
    for is_mini, (lo, hi) in ((1, (5.1, 7.9)), (0, (8.2, 9.6))):
        tab = (C.c_float * (nd * nm))()
        assert mini.c21_Nion_Conditional_table2d(D, lnMmin, lnMc, lnMc, s_c, dmin, dmax, lo, hi,
                                                 C.byref(sc), is_mini, 1, -40.0, 0, tab, nd, nm) == 0
        t = np.array(tab[:]).reshape(nd, nm)
        fn = mini.c21_Nion_ConditionalM_MINI if is_mini else mini.c21_Nion_ConditionalM
        for i, j in ((0, 0), (13, 49), (200, 25), (399, 7), (330, 0)):
            delta = dmin + np.float32(i) / (np.float32(nd) - 1.0) * (dmax - dmin)
            mt = 10 ** (lo + np.float32(j) / (np.float32(nm) - 1.0) * (hi - lo))
            direct = fn(D, lnMmin, lnMc, lnMc, s_c, float(delta), float(mt), C.byref(sc), 1)
            ln_direct = math.log(direct) if direct > 0 else -math.inf  # exp(-M_cond/M_acg) underflows
            assert t[i, j] == pytest.approx(max(ln_direct, -40.0), rel=3e-6, abs=3e-6), (i, j)
        # more collapse in denser regions (underdense rows: at high delta the small halos of the MCG
        # population merge away above the atomic threshold), less with a higher turnover mass
        assert np.all(np.diff(t[:150], axis=0) > 0)
        assert np.all(np.diff(t[:300], axis=1) < 0)  # (rows past the collapse threshold sit at -40)
    # the adaptive (QAG) method agrees with Gauss-Legendre for the MCG integrand
    for delta in (-0.5, 0.3):
        a = mini.c21_Nion_ConditionalM_MINI(D, lnMmin, lnMc, lnMc, s_c, delta, 2e6, C.byref(sc), 0)
        b = mini.c21_Nion_ConditionalM_MINI(D, lnMmin, lnMc, lnMc, s_c, delta, 2e6, C.byref(sc), 1)
        assert a == pytest.approx(b, rel=3e-3)


def test_redshift_tables_of_the_mini_population(mini):
    """The shared Gauss-Legendre rule behind Nion_z_table_MINI / SFRD_z_table_MINI against the
    adaptive Nion_General_MINI integrals with the constants evolved to each redshift."""
    lib = mini
    lib.c21_Nion_z_tables_mini.restype = C.c_int
    lib.c21_Nion_z_tables_mini.argtypes = [C.c_int, f64, f64, f64, C.POINTER(ScalingConsts), C.c_int,
                                           f64, f64, C.POINTER(f64), C.POINTER(f64)]
    lib.c21_scaling_consts_at_z.restype = ScalingConsts
    lib.c21_scaling_consts_at_z.argtypes = [f64, C.POINTER(ScalingConsts)]
    lib.c21_scaling_consts_sfr.restype = ScalingConsts
    lib.c21_scaling_consts_sfr.argtypes = [C.POINTER(ScalingConsts)]
    sc = ScalingConsts()
    assert lib.c21_set_scaling_constants(12.0, C.byref(sc)) == 0
    nz, nm = 40, 50
    z_min, z_w = 11.9, 0.55
    l10_min, l10_w = 5.0 - 9e-8, (10.0 - (5.0 - 9e-8)) / 49.0
    nion = (f64 * (nz * nm))()
    sfrd = (f64 * (nz * nm))()
    lnMmin, lnMmax = math.log(1e5), math.log(1e16)
    assert lib.c21_Nion_z_tables_mini(nz, z_min, z_w, lnMmin, C.byref(sc), nm, l10_min, l10_w,
                                      nion, sfrd) == 0
    tn = np.array(nion[:]).reshape(nz, nm)
    ts = np.array(sfrd[:]).reshape(nz, nm)
    for k, j in ((0, 0), (0, 49), (7, 12), (20, 30), (39, 5), (39, 44)):
        z = z_min + k * z_w
        mt = 10 ** (l10_min + j * l10_w)
        sc_z = lib.c21_scaling_consts_at_z(z, C.byref(sc))
        assert sc_z.acg_thresh < sc.acg_thresh or z <= 12.0
        want = lib.c21_Nion_General_MINI(z, lnMmin, lnMmax, mt, C.byref(sc_z))
        assert tn[k, j] == pytest.approx(want, rel=2e-5, abs=1e-300), (k, j)
        sc_s = lib.c21_scaling_consts_sfr(C.byref(sc_z))
        assert sc_s.fesc_7 == 1.0 and sc_s.alpha_esc == 0.0
        want_s = lib.c21_Nion_General_MINI(z, lnMmin, lnMmax, mt, C.byref(sc_s))
        assert ts[k, j] == pytest.approx(want_s, rel=2e-5, abs=1e-300), (k, j)
    assert np.all(np.diff(tn, axis=0) < 0) and np.all(np.diff(tn, axis=1) < 0)


def test_xray_table_of_both_populations_against_scipy(mini):
    """Xray_conditional_table_2D (interp_tables.c:497-560, hmf.c:482-509): the halo X-ray luminosity
    of both populations (metallicity from the summed star formation) against scipy quadrature of
    the Sheth-Tormen conditional mass function with the library's sigma(M)."""
    lib = mini
    sc = ScalingConsts()
    z = 12.0
    assert lib.c21_set_scaling_constants(z, C.byref(sc)) == 0
    D = lib.dicke(z)
    Mmin, Mcond = 1e5, lib.c21_RtoM(3.0)
    s_c = lib.c21_sigma_fast(Mcond)
    nd, nm = 400, 50
    dmin, dmax = -0.9, 1.2
    lo, hi = 5.0 - 9e-8, 10.0
    tab = (C.c_float * (nd * nm))()
    assert lib.c21_Nion_Conditional_table2d(D, math.log(Mmin), math.log(Mcond), math.log(Mcond), s_c,
                                            dmin, dmax, lo, hi, C.byref(sc), 2, 1, -50.0, 1, tab,
                                            nd, nm) == 0
    t = np.array(tab[:]).reshape(nd, nm)
    ob_om = lib._keep["cp"].OMb / lib._keep["cp"].OMm
    a, b, c = 0.73, 0.34, 0.81

    def cmf(lnM, delta):
        M = math.exp(lnM)
        s1, ds = lib.c21_sigma_fast(M), lib.dsigmasqdm_z0(M)
        if s1 < s_c:
            return 0.0
        diff = s1 * s1 - s_c * s_c
        dl = 1.686 / D
        terms, term = [1.0], 1.0
        for i in range(1, 6):
            term = term * (-diff) / i * (c - i + 1) / (s1 * s1)
            terms.append(term)
        p2 = b * (a * dl * dl / (s1 * s1)) ** (-c)
        factor = math.sqrt(a) * dl * (1 + p2 * sum(reversed(terms))) - delta / D
        barrier = math.sqrt(a) * dl * (1 + p2)
        return (-ds * factor * diff ** -1.5 * math.exp(-((barrier - delta / D) ** 2) * 0.5 / diff)
                / math.sqrt(2 * math.pi))

    def lx(lnM, mt):
        M = math.exp(lnM)

        def pl(norm, alpha, piv, lim):
            if (alpha > 0 and lnM > math.log(lim)) or (alpha < 0 and lnM < math.log(lim)):
                return -math.log(norm)
            return alpha * (lnM - piv * math.log(10))
        fs = math.exp(pl(sc.fstar_10, sc.alpha_star, 10, sc.Mlim_Fstar) - sc.mturn_a_nofb / M) * sc.fstar_10
        fm = math.exp(pl(sc.fstar_7, sc.alpha_star_mini, 7, sc.Mlim_Fstar_mini) - mt / M
                      - M / sc.acg_thresh) * sc.fstar_7
        stars, stars_m = M * fs * ob_om, M * fm * ob_om
        sfr, sfr_m = stars / (sc.t_star * sc.t_h), stars_m / (sc.t_star * sc.t_h)
        zscale = 10 ** (-0.056 * z + 0.064)
        st = 1.0
        if stars + stars_m > 0 and sfr + sfr_m > 0:
            M0 = 1.28825e10 * ((sfr + sfr_m) * 31556925.9747) ** 0.56
            st = (1 + ((stars + stars_m) / M0) ** -2.1) ** -0.148
        met = 1.23 * st * zscale
        ratio = 1.0 / ((met / 0.05) ** 0.0 + (met / 0.05) ** 0.64)  # USE_UPPER_STELLAR_TURNOVER
        return 31556925.9747 * (sfr * sc.l_x * ratio + sfr_m * sc.l_x_mini * ratio)

    for i, j in ((40, 5), (200, 20), (330, 35)):
        delta = dmin + np.float32(i) / (np.float32(nd) - 1.0) * (dmax - dmin)
        mt = float(np.float32(10 ** (lo + np.float32(j) / (np.float32(nm) - 1.0) * (hi - lo))))
        want, _ = integrate.quad(lambda x: lx(x, mt) * cmf(x, float(delta)), math.log(Mmin),
                                 math.log(Mcond), limit=400, epsrel=1e-8)
        assert t[i, j] == pytest.approx(math.log(want), abs=3e-3), (i, j)
    assert np.all(np.diff(t[:150], axis=0) > 0)
