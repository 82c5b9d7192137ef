"""Host tables of the spin temperature with the molecularly cooled population (USE_MINI_HALOS):
Pop-III Lyman-alpha and Lyman-Werner prefactors against a numpy restatement of the spectra, the
redshift x turnover tables against direct integrals, the tau_X = 1 frequencies and the per-shell
2-D SFRD tables.  reference: SpinTemperatureBox.c:374-499,930-1008, heating_helper_progs.c:
284-302,901-941,1094-1160, interp_tables.c:96-232,415-494."""
import ctypes as C
import math
from pathlib import Path

import numpy as np
import pytest

from oracle import ref_heating as RH
from test_host_heating import Tables, heat  # noqa: F401  (fixture)
from test_host_scalars import ScalingConsts

f64, f32, i32 = C.c_double, C.c_float, C.c_int
DATA = Path(__file__).parent / "golden" / "reference" / "_data"


@pytest.fixture()
def mini_heat(heat):
    keep = heat._keep
    sig = {"c21_ts_prepare_shells": (i32, [f32, f32, f32, C.c_void_p, C.c_void_p]),
           "c21_ts_prepare_tables": (i32, [f64, C.c_void_p, C.c_void_p]),
           "c21_ts_sfrd_tables": (i32, [C.POINTER(f64), C.POINTER(f64), C.c_void_p, C.c_void_p]),
           "c21_spectral_emissivity_lw": (f64, [f64, i32]),
           "c21_EvaluateNionTs_MINI": (f64, [f64, f64]), "c21_EvaluateSFRD_MINI": (f64, [f64, f64]),
           "c21_Nion_General_MINI": (f64, [f64] * 4 + [C.POINTER(ScalingConsts)]),
           "c21_Nion_ConditionalM_MINI": (f64, [f64] * 7 + [C.POINTER(ScalingConsts), i32]),
           "c21_set_scaling_constants": (i32, [f64, C.POINTER(ScalingConsts)]),
           "c21_scaling_consts_at_z": (ScalingConsts, [f64, C.POINTER(ScalingConsts)]),
           "c21_scaling_consts_sfr": (ScalingConsts, [C.POINTER(ScalingConsts)]),
           "c21_nu_tau_one_MINI": (f64, [f64] * 6 + [C.POINTER(i32)]),
           "c21_sigma_fast": (f64, [f64]), "dicke": (f64, [f64])}
    for name, (res, args) in sig.items():
        getattr(heat, name).restype = res
        getattr(heat, name).argtypes = args
    keep["ao"].USE_MINI_HALOS = True
    old = (keep["ap"].ALPHA_STAR_MINI, keep["ap"].F_STAR7_MINI)
    keep["ap"].ALPHA_STAR_MINI, keep["ap"].F_STAR7_MINI = 0.5, 10 ** -2.5
    yield heat
    keep["ao"].USE_MINI_HALOS = False
    keep["ap"].ALPHA_STAR_MINI, keep["ap"].F_STAR7_MINI = old


def lw_emissivity(sp, nu, pop):
    return sp.emissivity_lw(nu, pop)


def test_prepare_with_mini_halos(mini_heat, pkg):
    lib = mini_heat
    S = pkg.structs
    zp, prev_z, x_e_ave = 18.0, 18.76, 2.3e-4
    spec, tab = S.TsSpec(), Tables()
    assert lib.c21_ts_prepare_shells(zp, prev_z, zp, C.byref(spec), C.byref(tab)) == 0, pkg.last_error()
    n = tab.n_step
    assert spec.use_mini_halos == 1 and spec.sfr_scale_mini == pytest.approx(10 ** -2.5, rel=1e-6)
    assert spec.mturn_tab_min == 5.0 - 9e-8
    assert spec.xray_scale_mini == pytest.approx(lib._keep["ap"].L_X_MINI * RH.PC["s_per_yr"])
    # Lyman-alpha (Pop III) and Lyman-Werner prefactors of every shell from the spectra
    sp = RH.StellarSpectra(DATA / "stellar_spectra.dat")
    lw_edge = 2.70331197e15 / 3.288465e15
    for R in range(n):
        zpp = tab.zpp[R]
        ly2 = ly2m = lyn = lynm = lw = lwm = 0.0
        nup = RH.nu_n(2) * (1 + zpp) / (1 + zp)
        if zpp < RH.zmax(zp, 2):
            ly2 = RH.frecycle(2) * sp.emissivity(nup, 2)
            ly2m = RH.frecycle(2) * sp.emissivity(nup, 3)
            nul = max(nup, lw_edge)
            lw += lw_emissivity(sp, nul, 2)
            lwm += lw_emissivity(sp, nul, 3)
        for nn in range(23, 2, -1):
            if zpp > RH.zmax(zp, nn):
                continue
            nup = RH.nu_n(nn) * (1 + zpp) / (1 + zp)
            lyn += RH.frecycle(nn) * sp.emissivity(nup, 2)
            lynm += RH.frecycle(nn) * sp.emissivity(nup, 3)
            nul = max(nup, lw_edge)
            if nul >= RH.nu_n(nn + 1):
                continue
            lw += lw_emissivity(sp, nul, 2)
            lwm += lw_emissivity(sp, nul, 3)
        f = (1 + zp) ** 2 * (1 + zpp)
        if ly2 + lyn == 0:
            continue  # (the edge-of-horizon shell takes a fraction of its predecessor)
        assert spec.starlya_prefactor_mini[R] == pytest.approx(f * (ly2m + lynm), rel=3e-5)
        assert spec.lya_cont_prefactor_mini[R] == pytest.approx(f * ly2m, rel=3e-5)
        assert spec.lw_prefactor[R] == pytest.approx(f * lw, rel=3e-5)
        assert spec.lw_prefactor_mini[R] == pytest.approx(f * lwm, rel=3e-5)
    assert spec.lw_prefactor[0] > 0 and spec.lw_prefactor_mini[0] > 0

    # the global tables with a turnover mass per shell
    for R in range(n):
        tab.ave_log10_mturn[R] = 5.6 + 0.01 * R
    assert lib.c21_ts_prepare_tables(x_e_ave, C.byref(spec), C.byref(tab)) == 0, pkg.last_error()
    sc = ScalingConsts()
    assert lib.c21_set_scaling_constants(zp, C.byref(sc)) == 0
    lnMmin, lnMmax = math.log(1e5), math.log(1e16)
    z_top = tab.zpp[n - 1]  # the tables span [0.999 z', 1.001 z''_max]
    for z, l10 in ((18.3, 5.7), (0.5 * (zp + z_top), 6.4), (z_top, 5.2)):
        sc_z = lib.c21_scaling_consts_at_z(z, C.byref(sc))
        want = lib.c21_Nion_General_MINI(z, lnMmin, lnMmax, 10 ** l10, C.byref(sc_z))
        assert lib.c21_EvaluateNionTs_MINI(z, l10) == pytest.approx(want, rel=1e-2)  # bilinear table (0.1 dex bins)
        sc_s = lib.c21_scaling_consts_sfr(C.byref(sc_z))
        want = lib.c21_Nion_General_MINI(z, lnMmin, lnMmax, 10 ** l10, C.byref(sc_s))
        assert lib.c21_EvaluateSFRD_MINI(z, l10) == pytest.approx(want, rel=1e-2)
    for R in (0, 7, n - 1):
        assert spec.mean_sfr_zpp_mini[R] == lib.c21_EvaluateSFRD_MINI(tab.zpp[R], tab.ave_log10_mturn[R])
    ap = lib._keep["ap"]
    zeta = ap.F_STAR10 * ap.F_ESC10 * ap.POP2_ION
    zeta_m = ap.F_STAR7_MINI * ap.F_ESC7_MINI * ap.POP3_ION
    want_Q = 1 - (zeta * lib.c21_EvaluateNionTs(zp)
                  + zeta_m * lib.c21_EvaluateNionTs_MINI(zp, tab.ave_log10_mturn[0])) / (1 - x_e_ave)
    assert tab.Q_HI == pytest.approx(want_Q, rel=1e-8) and spec.no_light == 0  # (float products in C)
    # tau_X = 1 frequencies: the extra ionisation of the mini-halos can only lower them
    st = i32()
    for R in (3, 20, n - 1):
        with_m = tab.nu_tau_one[R]
        without = lib.c21_nu_tau_one(zp, tab.zpp[R], x_e_ave, zeta, C.byref(st))
        assert st.value == 0 and 0 < with_m <= without * 1.02
        assert with_m == lib.c21_nu_tau_one_MINI(zp, tab.zpp[R], x_e_ave, zeta, zeta_m,
                                                 tab.ave_log10_mturn[R], C.byref(st))

    # per-shell conditional tables of both populations
    lo = (f64 * 128)(*([-0.5] * 128))
    hi = (f64 * 128)(*([3.0] * 128))
    assert lib.c21_ts_sfrd_tables(lo, hi, C.byref(spec), C.byref(tab)) == 0, pkg.last_error()
    nd, nm = S.NDELTA_TABLE, S.NMTURN_TABLE
    t2 = np.ctypeslib.as_array(spec.ln_sfrd_tables_mini, (n, nd, nm))
    assert np.isfinite(t2).all() and t2.min() >= -50.0
    for R, i, j in ((0, 10, 3), (5, 200, 20), (n - 1, 399, 49), (12, 0, 0)):
        scR = ScalingConsts()
        assert lib.c21_set_scaling_constants(tab.zpp[R], C.byref(scR)) == 0
        scR = lib.c21_scaling_consts_sfr(C.byref(scR))
        g = tab.zpp_growth[R]
        dmin, dmax = -0.5 * g, 3.0 * g * 1.001
        delta = dmin + np.float32(i) / (np.float32(nd) - 1.0) * (dmax - dmin)
        mt = float(np.float32(10 ** (spec.mturn_tab_min + np.float32(j) / (np.float32(nm) - 1.0)
                                     * (10.0 - spec.mturn_tab_min))))
        lnMc = math.log(tab.M_max_R[R])
        direct = lib.c21_Nion_ConditionalM_MINI(
            lib.dicke(tab.zpp[R]), math.log(tab.M_min_R[R]), lnMc, lnMc,
            float(np.float32(lib.c21_sigma_fast(tab.M_max_R[R]))), float(delta), mt, C.byref(scR), 1)
        want = max(math.log(direct), -50.0) if direct > 0 else -50.0
        assert t2[R, i, j] == pytest.approx(want, rel=3e-6, abs=3e-6), (R, i, j)
