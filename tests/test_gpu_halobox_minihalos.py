"""GPU: ComputeHaloBox with USE_MINI_HALOS against the oracle -- the turnover grids (with
upstream's per-thread running maximum of the atomic turnover), the four-value deposit with 2-D
table lookups, and the entry point.  Tolerances as tests/test_gpu_halobox.py."""
import ctypes as C
import importlib
import math

import numpy as np
import pytest

import halobox_mini_helpers as HM
from test_gpu_halobox import api, compare  # noqa: F401  (fixture)
from test_oracle_halobox import random_ics

pytestmark = pytest.mark.gpu
S = importlib.import_module("21cmfast_amd.structs")


@pytest.mark.parametrize("n,lpt2,vscale,device,xray", [(16, 1, 1.0, False, True), (32, 1, 6.0, True, True),
                                                       (24, 0, 20.0, True, False), (33, 1, 3.0, False, True)])
def test_deposit_matches_oracle(api, oracle, n, lpt2, vscale, device, xray):
    spec = HM.mini_spec(n, lpt2=lpt2, xray=xray)
    ics = random_ics(n, n, False, seed=n, vscale=vscale)
    ref = oracle.halobox_grids(spec, ics, with_whalo=True, with_xray=xray)
    if device:
        import torch

        ics = {k: torch.from_numpy(v).cuda() for k, v in ics.items()}
    got = api.halobox_grids(spec, ics, with_whalo=True, with_xray=xray)
    assert set(got) == set(ref) and "halo_sfr_mini" in got
    compare(got, ref)
    assert ref["halo_sfr_mini"].max() > 0


@pytest.mark.parametrize("threads", [1, 5, 16])
def test_turnovers_match_oracle(api, oracle, threads):
    shape = (20, 20, 28)
    spec, g12, zre, j21, vcb = HM.turnover_inputs(shape)
    for v in (vcb, None):
        ra, rm, rave = oracle.halobox_turnovers(spec, 1e5, 1, threads, g12, zre, j21, v)
        a, m, ave = api.halobox_turnovers(spec, 1e5, 1, threads, g12, zre, j21, v)
        np.testing.assert_allclose(a, ra, rtol=3e-7)
        np.testing.assert_allclose(m, rm, rtol=3e-7)
        assert ave == pytest.approx(rave, rel=1e-7)
    import torch
    dev = lambda x: torch.from_numpy(x).cuda()  # noqa: E731
    a, m, ave = api.halobox_turnovers(spec, 1e5, 1, threads, dev(g12), dev(zre), dev(j21), dev(vcb))
    ra, rm, rave = oracle.halobox_turnovers(spec, 1e5, 1, threads, g12, zre, j21, vcb)
    np.testing.assert_allclose(a.cpu().numpy(), ra, rtol=3e-7)
    # above Z_HEAT_MAX: no inputs needed
    a0, m0, _ = api.halobox_turnovers(spec, 1e5, 0, threads, None, None, None, vcb, like=vcb)
    r0 = oracle.halobox_turnovers(spec, 1e5, 0, threads, None, None, None, vcb, shape=shape)
    np.testing.assert_allclose(a0, r0[0], rtol=3e-7)
    np.testing.assert_allclose(m0, r0[1], rtol=3e-7)


def test_high_resolution_sources_are_refused(api):
    from test_oracle_halobox import halobox_spec, make_tables
    spec = HM.add_minis(halobox_spec(16, 32, True, make_tables()), 16)
    ics = random_ics(16, 32, True, seed=1)
    with pytest.raises(RuntimeError, match="PERTURB_ON_HIGH_RES"):
        api.halobox_grids(spec, ics)


@pytest.mark.parametrize("use_ts,threads", [(True, 4), (False, 1)])
def test_compute_halo_box_with_mini_halos(gpu_lib, oracle, tmp_path, use_ts, threads):
    """ComputeHaloBox, L-INTEGRAL + USE_MINI_HALOS: turnover grids from the previous boxes (running
    maximum over N_THREADS shares), table ranges from them, the 2-D tables of the host side (checked
    in tests/test_host_minihalos.py) and the deposit -- against the oracle fed with the same tables."""
    from test_gpu_abi import Session, fptr
    from test_host_scalars import ScalingConsts
    import test_host_minihalos as THM

    lib = gpu_lib
    n = 24
    ses = Session(lib, tmp_path, HII_DIM=n, DIM=2 * n, SOURCE_MODEL=2, USE_MINI_HALOS=True,
                  ALPHA_STAR_MINI=0.5, USE_TS_FLUCT=use_ts, M_TURN=10 ** 5.0, N_THREADS=threads,
                  Z_HEAT_MAX=30.0, V_CB_MODEL=3, RECOMB_MODEL=2)
    THM._bind(lib)
    z = 10.0
    shape = (n, n, n)
    ics = random_ics(n, 2 * n, False, seed=9)
    ics["lowres_density"] = (ics["lowres_density"] * 0.5).astype(np.float32)
    _, g12, zre, j21, _ = HM.turnover_inputs(shape)
    fields = ["n_ion", "halo_sfr", "halo_sfr_mini", "whalo_sfr"] + (["halo_xray"] if use_ts else [])
    out = {k: np.zeros(shape, np.float32) for k in fields}
    hb = S.HaloBoxStruct(**{k: fptr(v) for k, v in out.items()})
    icss = S.InitialConditionsStruct(**{k: fptr(v) for k, v in ics.items()})
    pts = S.TsBoxStruct(J_21_LW=fptr(j21))
    pion = S.IonizedBoxStruct(ionisation_rate_G12=fptr(g12), z_reion=fptr(zre))
    lib.ComputeHaloBox.argtypes = [C.c_double] + [C.c_void_p] * 5
    st = lib.ComputeHaloBox(z, C.byref(icss), None, C.byref(pts), C.byref(pion), C.byref(hb))
    assert st == 0, lib.c21cm_last_error()

    f64 = C.c_double
    lib.sigma_z0.restype = f64
    lib.sigma_z0.argtypes = [f64]
    lib.c21_Nion_Conditional_table.restype = C.c_int
    lib.c21_Nion_Conditional_table.argtypes = [f64] * 8 + [C.POINTER(ScalingConsts), C.c_int, f64,
                                                           C.POINTER(C.c_float), C.c_int]
    lib.c21_scaling_consts_sfr.restype = ScalingConsts
    lib.c21_scaling_consts_sfr.argtypes = [C.POINTER(ScalingConsts)]
    sc = ScalingConsts()
    assert lib.c21_set_scaling_constants(z, C.byref(sc)) == 0
    sc_sfrd = lib.c21_scaling_consts_sfr(C.byref(sc))
    ms = S.MturnSpec(hii_dim=n, hii_dim_z=n, redshift=z, mturn_a_nofb=sc.mturn_a_nofb,
                     vcb_const=sc.vcb_const, A_LW=ses.ap.A_LW, BETA_LW=ses.ap.BETA_LW,
                     A_VCB=ses.ap.A_VCB, BETA_VCB=ses.ap.BETA_VCB,
                     sigma_vcb=ses.ct.V_CB_AVG * math.sqrt(3 * math.pi / 8))
    mta, mtm, ave = oracle.halobox_turnovers(ms, ses.ap.M_TURN, 1, threads, g12, zre, j21, None)
    assert hb.log10_Mcrit_ACG_ave == pytest.approx(ave[0], rel=1e-7)
    assert hb.log10_Mcrit_MCG_ave == pytest.approx(ave[1], rel=1e-7)
    D = lib.dicke(z)
    d = ics["lowres_density"].astype(np.float64) * D
    dmin, dmax = min(0.0, d.min()) * 1.001, max(0.0, d.max()) * 1.001
    M_min = lib.c21_minimum_source_mass(z)
    assert M_min == 1e5
    M_cell = lib.c21_rhocrit() * ses.cp.OMm * ses.so.BOX_LEN ** 3 / n ** 3
    lnMmin, lnMmax, lnMc, sig = math.log(M_min), math.log(1e16), math.log(M_cell), lib.sigma_z0(M_cell)
    a_lo, a_hi = min(16.0, float(mta.min())) * 0.999, max(5.0, float(mta.max())) * 1.001
    m_lo, m_hi = min(16.0, float(mtm.min())) * 0.999, max(5.0, float(mtm.max())) * 1.001
    nd, nm = S.NDELTA_TABLE, S.NMTURN_TABLE
    t2 = [np.zeros((nd + 1, nm), np.float32) for _ in range(4)]
    fp = lambda a: a.ctypes.data_as(S.c_float_p)  # noqa: E731
    fixed = (5.0 - 9e-8, 10.0)
    sig_f = float(np.float32(sig))
    calls = [(t2[0], sig, a_lo, a_hi, sc, 0, -40.0, 0), (t2[1], sig, m_lo, m_hi, sc, 1, -40.0, 0),
             (t2[2], sig_f, *fixed, sc_sfrd, 1, -50.0, 1)] + (
                 [(t2[3], sig, *fixed, sc, 2, -50.0, 1)] if use_ts else [])
    for tab, sg, lo, hi, consts, kind, floor, fl in calls:
        assert lib.c21_Nion_Conditional_table2d(D, lnMmin, lnMmax, lnMc, sg, dmin, dmax, lo, hi,
                                                C.byref(consts), kind, 1, floor, fl, fp(tab), nd, nm) == 0
    t1 = (C.c_float * nd)()
    assert lib.c21_Nion_Conditional_table(D, lnMmin, lnMmax, lnMc, sig, dmin, dmax, sc_sfrd.mturn_a_nofb,
                                          C.byref(sc_sfrd), 1, -50.0, t1, nd) == 0
    rho_b = lib.c21_rhocrit() * ses.cp.OMb
    spec = S.HaloBoxSpec(dim=2 * n, dim_z=2 * n, hii_dim=n, hii_dim_z=n, box_len=ses.so.BOX_LEN,
                         box_len_z=ses.so.BOX_LEN, perturb_on_high_res=0, lpt2=1, growth_factor=D,
                         init_growth_factor=lib.dicke(ses.so.INITIAL_REDSHIFT), tab_min=dmin,
                         tab_width=(dmax - dmin) / (nd - 1.0), ln_nion_table=C.cast(t1, S.c_float_p),
                         ln_sfrd_table=C.cast(t1, S.c_float_p),
                         prefactor_nion=rho_b * sc.fstar_10 * sc.fesc_10 * sc.pop2_ion,
                         prefactor_sfr=rho_b * sc.fstar_10 / sc.t_star / sc.t_h,
                         prefactor_wsfr=1 / sc.t_h / sc.t_star,
                         prefactor_xray=lib.c21_rhocrit() * ses.cp.OMm)
    spec.use_mini_halos = 1
    spec.log10_mturn_acg, spec.log10_mturn_mcg = fp(mta), fp(mtm)
    spec.ln_nion_table2d, spec.ln_nion_mini_table2d, spec.ln_sfrd_mini_table2d = fp(t2[0]), fp(t2[1]), fp(t2[2])
    if use_ts:
        spec.ln_xray_table2d = fp(t2[3])
    spec.mta_min, spec.mta_width = a_lo, (a_hi - a_lo) / (nm - 1.0)
    spec.mtm_min, spec.mtm_width = m_lo, (m_hi - m_lo) / (nm - 1.0)
    spec.mt_fixed_min, spec.mt_fixed_width = fixed[0], (fixed[1] - fixed[0]) / (nm - 1.0)
    spec.prefactor_sfr_mini = rho_b * sc.fstar_7 / sc.t_star / sc.t_h
    spec.prefactor_nion_mini = rho_b * sc.fstar_7 * sc.fesc_7 * sc.pop3_ion
    ref = oracle.halobox_grids(spec, ics, with_whalo=True, with_xray=use_ts)
    compare(out, ref)
    assert out["halo_sfr_mini"].max() > 0 and (out["n_ion"] > 0).all()
    del ses
