"""GPU parity of the mini-halo (USE_MINI_HALOS, E-INTEGRAL) ionisation path against the oracle:
turnover-mass boxes, the per-radius f_coll history of both populations, the two-population barrier
with and without recombinations / x_e grid.  Tolerances as tests/test_gpu_ionize.py; the history
grids f = f_prev + f(z) - f(z_prev) cancel, so they are compared with an absolute floor of 2e-7
(float32 transforms move the filtered inputs by ~1e-6 relative).
Reference behaviour: IonisationBox.c:403-457, 715-761, 838-936, 1068-1200."""
import importlib
import math

import numpy as np
import pytest

import mini_helpers as H
from test_gpu_ionize import api, compare  # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu

S = importlib.import_module("21cmfast_amd.structs")
W = importlib.import_module("21cmfast_amd.workloads")


def run_device(api, spec, density, mini, device_resident, **kw):
    if device_resident:
        import torch

        dev = lambda a: None if a is None else torch.from_numpy(a).cuda()  # noqa: E731
        buf, box, rep = api.ionize_grids(spec, dev(density), mini={k: dev(v) for k, v in mini.items()},
                                         **{k: dev(v) for k, v in kw.items()})
        torch.cuda.synchronize()
        host = lambda a: None if a is None else a.cpu().numpy()  # noqa: E731
    else:
        buf, box, rep = api.ionize_grids(spec, density, mini=mini, **kw)
        host = lambda a: a  # noqa: E731
    out = {k: host(getattr(buf, k)) for k in
           ("neutral_fraction", "z_reion", "kinetic_temperature", "unnormalised_nion",
            "unnormalised_nion_mini", "ionisation_rate_G12", "mean_free_path",
            "cumulative_recombinations")}
    out["report"] = rep
    out["mean_f_coll"] = box.mean_f_coll
    out["mean_f_coll_MINI"] = box.mean_f_coll_MINI
    return out


def compare_mini(got, ref, spec, **kw):
    compare(got, ref, spec, **kw)
    n = spec.n_radii
    np.testing.assert_allclose(np.array(got["report"].f_coll_grid_mean_mini[:n]),
                               np.array(ref["report"].f_coll_grid_mean_mini[:n]), rtol=1e-5)
    for k in ("unnormalised_nion", "unnormalised_nion_mini"):
        assert got[k].shape == (n,) + got["neutral_fraction"].shape
        np.testing.assert_allclose(got[k], ref[k], rtol=2e-4, atol=2e-7)
    assert got["mean_f_coll"] == pytest.approx(ref["mean_f_coll"], rel=1e-5)
    assert got["mean_f_coll_MINI"] == pytest.approx(ref["mean_f_coll_MINI"], rel=1e-5)


@pytest.mark.parametrize("shape,device_resident", [((32, 32, 32), False), ((64, 64, 64), True),
                                                   ((35, 35, 35), False), ((32, 32, 64), True)])
@pytest.mark.parametrize("need_prev", [1, 0])
def test_two_population_parity(api, oracle, shape, device_resident, need_prev):
    n, nz = shape[0], shape[2]
    spec = H.mini_spec(n, nz=nz, need_prev=need_prev, r_bubble_max=12.0)
    density, mini = H.mini_inputs(shape, spec.n_radii, history=bool(need_prev))
    ref = oracle.ionize_grids(spec, density, mini=mini)
    got = run_device(api, spec, density, mini, device_resident)
    compare_mini(got, ref, spec)
    assert 0.02 < (ref["neutral_fraction"] == 0).mean() < 0.98


def test_with_xe_grid_and_previous_reionisation(api, oracle):
    n = 48
    shape = (n, n, n)
    spec = H.mini_spec(n, need_prev=1, r_bubble_max=12.0, use_ts_fluct=1, first_snapshot=0)
    density, mini = H.mini_inputs(shape, spec.n_radii)
    rng = np.random.default_rng(21)
    xe = (-0.05 + 0.5 * rng.random(shape) ** 2).astype(np.float32)
    Tn = (8.0 + 4.0 * rng.random(shape)).astype(np.float32)
    pz = np.where(rng.random(shape) < 0.2, 11.0, -1.0).astype(np.float32)
    kw = dict(xe=xe, Tneutral=Tn, prev_z_reion=pz)
    ref = oracle.ionize_grids(spec, density, mini=mini, **kw)
    got = run_device(api, spec, density, mini, True, **kw)
    compare_mini(got, ref, spec)


@pytest.mark.parametrize("model,cell_recomb", [(2, 1), (2, 0), (1, 1)])
def test_recombination_models(api, oracle, model, cell_recomb):
    n = 40
    shape = (n, n, n)
    spec = H.mini_spec(n, need_prev=1, r_bubble_max=10.0, recomb_model=model)
    spec.cell_recomb = cell_recomb
    density, mini = H.mini_inputs(shape, spec.n_radii)
    rng = np.random.default_rng(9)
    if model == 2:
        prev_nrec = (0.3 * rng.random(shape)).astype(np.float32)
    else:
        prev_nrec = np.full((1, 1, 1), 0.15, np.float32)
    pz = np.where(rng.random(shape) < 0.1, 11.5, -1.0).astype(np.float32)
    kw = dict(prev_nrec=prev_nrec, prev_z_reion=pz)
    ref = oracle.ionize_grids(spec, density, mini=mini, **kw)
    got = run_device(api, spec, density, mini, model == 2, **kw)
    crossed = (got["mean_free_path"] > 0, ref["mean_free_path"] > 0)
    compare_mini(got, ref, spec, flags=crossed)
    same = crossed[0] == crossed[1]
    np.testing.assert_allclose(got["ionisation_rate_G12"][same], ref["ionisation_rate_G12"][same],
                               rtol=2e-4, atol=1e-7)
    np.testing.assert_array_equal(got["mean_free_path"][same], ref["mean_free_path"][same])
    np.testing.assert_allclose(got["cumulative_recombinations"].ravel()[0],
                               ref["cumulative_recombinations"].ravel()[0], rtol=2e-4)


def test_mturn_grids_parity(api, oracle):
    shape = (24, 24, 40)
    rng = np.random.default_rng(5)
    spec = S.MturnSpec(hii_dim=24, hii_dim_z=40, first_snapshot=0, redshift=11.0,
                       mturn_a_nofb=2.0e8, mturn_m_nofb=8.0e5, vcb_const=21.0, A_LW=2.0,
                       BETA_LW=0.6, A_VCB=1.0, BETA_VCB=1.8,
                       sigma_vcb=25.86 * math.sqrt(3 * math.pi / 8))
    g12 = (0.3 * rng.random(shape)).astype(np.float32)
    zre = np.where(rng.random(shape) < 0.4, 10.5 + 4 * rng.random(shape), -1.0).astype(np.float32)
    j21 = (0.5 * rng.random(shape) ** 2).astype(np.float32)
    vcb = (30 * rng.random(shape)).astype(np.float32)
    for v in (vcb, None):
        for first in (0, 1):
            spec.first_snapshot = first
            ra, rm, rave_a, rave_m = oracle.mturn_grids(spec, g12, zre, j21, v)
            a, m, ave_a, ave_m = api.mturn_grids(spec, g12, zre, j21, v)
            np.testing.assert_allclose(a, ra, rtol=3e-7)
            np.testing.assert_allclose(m, rm, rtol=3e-7)
            assert ave_a == pytest.approx(rave_a, rel=1e-7)
            assert ave_m == pytest.approx(rave_m, rel=1e-7)
    # device-resident arrays
    import torch
    dev = lambda x: torch.from_numpy(x).cuda()  # noqa: E731
    spec.first_snapshot = 0
    a, m, ave_a, _ = api.mturn_grids(spec, dev(g12), dev(zre), dev(j21), dev(vcb))
    ra, rm, rave_a, _ = oracle.mturn_grids(spec, g12, zre, j21, vcb)
    np.testing.assert_allclose(a.cpu().numpy(), ra, rtol=3e-7)
    assert ave_a == pytest.approx(rave_a, rel=1e-7)
    # a negative Lyman-Werner background makes the threshold NaN: refused (IonisationBox.c:425)
    bad = j21.copy()
    bad[3, 4, 5] = -1.0
    with pytest.raises(RuntimeError):
        api.mturn_grids(spec, g12, zre, bad, vcb)


def test_refusals(api):
    n = 16
    spec = H.mini_spec(n, r_bubble_max=6.0)
    density, mini = H.mini_inputs((n, n, n), spec.n_radii)
    spec.fcoll_mode = W.FCOLL_ERFC  # mini-halos live on the E-INTEGRAL tables only
    with pytest.raises(RuntimeError, match="E-INTEGRAL"):
        api.ionize_grids(spec, density, mini=mini)
    spec = H.mini_spec(n, r_bubble_max=6.0)
    import torch
    fc = torch.zeros(n ** 3, dtype=torch.uint8, device="cuda")
    with pytest.raises(RuntimeError):
        api.ionize_shard_radii(spec, 0, 2, fc, density)


def test_two_population_parity_with_mean_fix(api, oracle):
    """fix_mean = 1 (what ComputeIonizedBox sets for the Eulerian models): both populations are
    rescaled to their global means (IonisationBox.c:1022-1027)."""
    n = 40
    spec = H.mini_spec(n, need_prev=1, r_bubble_max=10.0)
    spec.fix_mean = 1
    spec.mean_f_coll, spec.mean_f_coll_mini = 0.014, 0.002
    density, mini = H.mini_inputs((n, n, n), spec.n_radii)
    ref = oracle.ionize_grids(spec, density, mini=mini)
    got = run_device(api, spec, density, mini, True)
    compare_mini(got, ref, spec)
    assert got["mean_f_coll_MINI"] == 0.002 and 0.02 < (ref["neutral_fraction"] == 0).mean() < 0.98


def test_compute_ionized_box_with_mini_halos(gpu_lib, oracle, tmp_path):
    """ComputeIonizedBox with USE_MINI_HALOS over two snapshots (first: no history; second: the
    trapezoidal history, reionisation feedback from the first box's Gamma_12 / z_reion) against the
    oracle fed with the host scalars and tables that tests/test_host_minihalos.py checks."""
    import ctypes as C

    from test_gpu_abi import Session, fptr
    from test_host_scalars import ScalingConsts
    import test_host_minihalos as HM

    lib = gpu_lib
    n = 32
    shape = (n, n, n)
    ses = Session(lib, tmp_path, HII_DIM=n, SOURCE_MODEL=1, HII_FILTER=1, USE_EXP_FILTER=False,
                  R_BUBBLE_MAX=10.0, USE_MINI_HALOS=True, RECOMB_MODEL=2, CELL_RECOMB=True,
                  ALPHA_STAR_MINI=0.5, F_STAR7_MINI=10 ** -2.0, F_ESC7_MINI=10 ** -1.5,
                  V_CB_MODEL=3)
    HM._bind(lib)
    f64 = C.c_double
    lib.c21_dtdz.restype = f64
    lib.c21_dtdz.argtypes = [C.c_float]
    lib.c21_nb0.restype = f64
    lib.c21_rr_tables.restype = C.c_int
    y, cc = C.POINTER(f64)(), C.POINTER(f64)()
    assert lib.c21_rr_tables(C.byref(y), C.byref(cc)) == 0

    rng = np.random.default_rng(17)
    dens = {12.5: W.density_field_numpy(n, seed=5, sigma=0.6)}
    dens[14.0] = (0.9 * dens[12.5]).astype(np.float32)
    j21 = {z: ((3.0 if z == 12.5 else 1.5) * rng.random(shape) ** 2).astype(np.float32) for z in dens}

    def new_box(n_radii):
        return {"neutral_fraction": np.ones(shape, np.float32), "z_reion": np.zeros(shape, np.float32),
                "kinetic_temperature": np.zeros(shape, np.float32),
                "ionisation_rate_G12": np.zeros(shape, np.float32),
                "mean_free_path": np.zeros(shape, np.float32),
                "cumulative_recombinations": np.zeros(shape, np.float32),
                "unnormalised_nion": np.zeros((n_radii,) + shape, np.float32),
                "unnormalised_nion_mini": np.zeros((n_radii,) + shape, np.float32)}

    n_radii = W.ionize_spec(n, box_len=ses.so.BOX_LEN, mode=W.FCOLL_TABLE_EXP,
                            r_bubble_max=ses.ap.R_BUBBLE_MAX).n_radii
    LN_MAX = math.log(1e16)

    def oracle_snapshot(z, prev_z, prev_arr, prev_means, prev_density):
        first = prev_z < 1
        spec = W.ionize_spec(n, box_len=ses.so.BOX_LEN, mode=W.FCOLL_TABLE_EXP,
                             r_bubble_max=ses.ap.R_BUBBLE_MAX, redshift=z)
        sc = ScalingConsts()
        assert lib.c21_set_scaling_constants(z, C.byref(sc)) == 0
        M_min = lib.c21_minimum_source_mass(z)
        assert M_min == 1e5  # hmf.c:1327-1329
        lnMmin = math.log(M_min)
        spec.hii_filter, spec.stars_filter = 1, 1
        spec.r_lowest = 0
        spec.sigma_minmass = lib.c21_sigma_fast(M_min)
        spec.growth_factor = lib.dicke(z)
        spec.TK_nofluct = lib.c21_T_RECFAST(z)
        spec.adia_TK_term = float(np.float32(0.58 - 0.006 * (np.float32(z) - 10.0)))
        spec.T_re = ses.ap.T_RE
        spec.rhocrit_omb = lib.c21_rhocrit() * ses.cp.OMb
        spec.mass_dep_zeta, spec.fix_mean = 1, 1
        spec.first_snapshot = int(first)
        spec.recomb_model, spec.cell_recomb = 2, 1
        spec.rr_y, spec.rr_c = y, cc
        spec.dz = (1 + z) * (ses.so.ZPRIME_STEP_FACTOR - 1) if first else prev_z - z
        spec.fabs_dtdz = abs(lib.c21_dtdz(z)) / 1e15
        zeta = sc.pop2_ion * sc.fstar_10 * sc.fesc_10
        zeta_m = sc.pop3_ion * sc.fstar_7 * sc.fesc_7
        spec.ion_eff_factor = zeta
        a_uvb = ses.ap.ALPHA_UVB
        spec.gamma_prefactor = ((1 + z) ** 2 * 3.08567758e24 * 6.3e-18 * a_uvb / (a_uvb + 2.75)
                                * lib.c21_nb0() * zeta / 1e-12 / (sc.t_h * sc.t_star))
        # calculate_mcrit_boxes
        ms = S.MturnSpec(hii_dim=n, hii_dim_z=n, first_snapshot=int(first), redshift=z,
                         mturn_a_nofb=sc.mturn_a_nofb, mturn_m_nofb=sc.mturn_m_nofb,
                         vcb_const=sc.vcb_const, A_LW=ses.ap.A_LW, BETA_LW=ses.ap.BETA_LW,
                         A_VCB=ses.ap.A_VCB, BETA_VCB=ses.ap.BETA_VCB,
                         sigma_vcb=ses.ct.V_CB_AVG * math.sqrt(3 * math.pi / 8))
        assert sc.vcb_const == ses.ap.V_CB_AVG_DEBUG  # V_CB_MODEL = AVG-DEBUG
        mta, mtm, ave_a, ave_m = oracle.mturn_grids(ms, prev_arr["ionisation_rate_G12"],
                                                    prev_arr["z_reion"], j21[z])
        Mt_a, Mt_m = 10 ** ave_a, 10 ** ave_m
        # set_mean_fcoll
        pa, pm = prev_means
        f_a = lib.c21_Nion_General(z, lnMmin, LN_MAX, Mt_a, C.byref(sc))
        f_m = lib.c21_Nion_General_MINI(z, lnMmin, LN_MAX, Mt_m, C.byref(sc))
        if not pa * zeta < 1e-4:
            f_a = pa + f_a - lib.c21_Nion_General(prev_z, lnMmin, LN_MAX, Mt_a, C.byref(sc))
        if not pm * zeta < 1e-4:
            f_m = pm + f_m - lib.c21_Nion_General_MINI(prev_z, lnMmin, LN_MAX, Mt_m, C.byref(sc))
        spec.mean_f_coll, spec.mean_f_coll_mini = f_a, f_m
        spec.f_limit_acg = lib.c21_Nion_General(ses.so.Z_HEAT_MAX, lnMmin, LN_MAX, Mt_a, C.byref(sc))
        spec.f_limit_mcg = lib.c21_Nion_General_MINI(ses.so.Z_HEAT_MAX, lnMmin, LN_MAX, Mt_m,
                                                     C.byref(sc))
        spec.use_mini_halos = 1
        spec.need_prev_ion = int(pm * zeta_m + pa * zeta > 1e-4)
        spec.ion_eff_factor_mini = zeta_m
        spec.gamma_prefactor_mini = spec.gamma_prefactor * zeta_m / zeta
        D_prev = lib.dicke(prev_z)

        def table2d_fn(r_index, prev, dmin, dmax, amin, amax, mmin, mmax, tab_a, tab_m, user):
            M_R = lib.c21_RtoM(spec.R[r_index])
            D = D_prev if prev else spec.growth_factor
            args = (D, lnMmin, math.log(M_R), math.log(M_R), lib.c21_sigma_fast(M_R), dmin, dmax)
            st = lib.c21_Nion_Conditional_table2d(*args, amin, amax, C.byref(sc), 0, 1, -40.0, 0,
                                                  tab_a, S.NDELTA_TABLE, S.NMTURN_TABLE)
            return st or lib.c21_Nion_Conditional_table2d(*args, mmin, mmax, C.byref(sc), 1, 1,
                                                          -40.0, 0, tab_m, S.NDELTA_TABLE,
                                                          S.NMTURN_TABLE)
        cb = S.TABLE2D_FN(table2d_fn)
        spec.table2d_fn = cb
        mini = dict(prev_density=prev_density, log10_mturn_acg=mta, log10_mturn_mcg=mtm,
                    prev_nion=prev_arr["unnormalised_nion"],
                    prev_nion_mini=prev_arr["unnormalised_nion_mini"])
        ref = oracle.ionize_grids(spec, dens[z], mini=mini, prev_z_reion=prev_arr["z_reion"],
                                  prev_nrec=prev_arr["cumulative_recombinations"])
        ref["ave"] = (ave_a, ave_m)
        ref["spec_means"] = (f_a, f_m)
        return ref

    def abi_snapshot(z, prev_z, prev_arr, prev_means, prev_density):
        arr = new_box(n_radii)
        box = S.IonizedBoxStruct(**{k: fptr(v) for k, v in arr.items()})
        prevs = S.IonizedBoxStruct(**{k: fptr(v) for k, v in prev_arr.items()})
        prevs.mean_f_coll, prevs.mean_f_coll_MINI = prev_means
        pf = S.PerturbedFieldStruct(density=fptr(dens[z]))
        ppf = S.PerturbedFieldStruct(density=fptr(prev_density))
        ts = S.TsBoxStruct(J_21_LW=fptr(j21[z]))
        hb, ics = S.HaloBoxStruct(), S.InitialConditionsStruct()
        st = lib.ComputeIonizedBox(z, prev_z, C.byref(pf), C.byref(ppf), C.byref(prevs),
                                   C.byref(ts), C.byref(hb), C.byref(ics), C.byref(box))
        assert st == 0, lib.c21cm_last_error()
        arr["means"] = (box.mean_f_coll, box.mean_f_coll_MINI)
        arr["ave"] = (box.log10_Mturnover_ave, box.log10_Mturnover_MINI_ave)
        return arr

    def check(got, ref):
        flag_g, flag_r = got["mean_free_path"] > 0, ref["mean_free_path"] > 0
        assert np.mean(flag_g != flag_r) <= 2e-4
        same = flag_g == flag_r
        for k in ("neutral_fraction", "ionisation_rate_G12", "mean_free_path",
                  "cumulative_recombinations", "z_reion"):
            np.testing.assert_allclose(got[k][same], ref[k][same], rtol=2e-4, atol=5e-6, err_msg=k)
        for k in ("unnormalised_nion", "unnormalised_nion_mini"):
            # (the real tables fall to the -40 floor above the collapse threshold: a cell whose
            # float32-filtered delta sits on that cliff moves with the transform's round-off)
            off = ~np.isclose(got[k], ref[k], rtol=2e-4, atol=2e-7)
            assert off.mean() <= 1e-5, (k, off.sum())
            np.testing.assert_allclose(got[k], ref[k], rtol=2e-2, atol=2e-7, err_msg=k)
        assert got["ave"] == pytest.approx(ref["ave"], rel=1e-7)
        assert got["means"] == pytest.approx(ref["spec_means"], rel=1e-10)  # fix_mean: kept
        return flag_r.mean()

    # snapshot 1: no previous box (prev_redshift = 0)
    zero_prev = new_box(n_radii)
    pd1 = np.zeros(shape, np.float32)
    got1 = abi_snapshot(14.0, 0.0, zero_prev, (0.3, 0.3), pd1)  # means are reset to 0 (:389-390)
    assert (pd1 == -1.5).all() and (zero_prev["z_reion"] == -1).all()
    ref_prev = new_box(n_radii)
    ref1 = oracle_snapshot(14.0, 0.0, ref_prev, (0.0, 0.0), np.full(shape, -1.5, np.float32))
    frac1 = check(got1, ref1)
    assert got1["unnormalised_nion_mini"].max() > 0
    # snapshot 2: history + feedback from snapshot 1
    means1 = got1.pop("means")
    got1.pop("ave")
    got2 = abi_snapshot(12.5, 14.0, got1, means1, dens[14.0].copy())
    ref1_arr = {k: ref1[k] for k in new_box(1)}
    ref2 = oracle_snapshot(12.5, 14.0, ref1_arr, ref1["spec_means"], dens[14.0])
    frac2 = check(got2, ref2)
    assert 0.01 < frac1 < frac2 < 0.98
    # reionisation feedback of snapshot 1's ionised cells (2.5e7 Msun here) lifts the MCG turnover
    # above its Lyman-Werner value there; the ACG turnover stays at M_TURN
    ms_nofb = oracle_snapshot(12.5, 14.0, new_box(n_radii), ref1["spec_means"], dens[14.0])["ave"]
    assert got2["ave"][0] == pytest.approx(math.log10(ses.ap.M_TURN), rel=1e-7)
    assert got2["ave"][1] > ms_nofb[1] > 6.0
    # the trapezoid used the history: means differ from the plain integrals
    assert got2["means"][0] != pytest.approx(got1["unnormalised_nion"][0].mean(), rel=1e-3)
    lib.free_MHR.restype = None
    lib.free_MHR()


@pytest.mark.parametrize("n,recomb,device_resident", [(64, 0, True), (50, 0, False), (40, 2, True)])
def test_lagrangian_grids_with_mini_halo_floor(api, oracle, n, recomb, device_resident):
    """Lagrangian source grids under USE_MINI_HALOS: the molecularly cooled photons are inside
    HaloBox.n_ion, the barrier only gains the floor f_limit_mcg x ion_eff_factor_mini (= 1 for
    these models) and box->mean_f_coll_MINI returns that floor (IonisationBox.c:49-50,1068-1082,
    1570-1573).  The floor is exaggerated here so that it decides cells."""
    from test_gpu_ionize import run_device as run_plain
    if recomb:
        from recomb_helpers import inputs, recomb_spec
        spec = recomb_spec(n, model=recomb)
        d = inputs((n, n, n))
        kw = dict(prev_nrec=d["prev_nrec"], whalo_sfr=d["whalo_sfr"], prev_z_reion=d["prev_z_reion"])
        density, n_ion = d["density"], d["n_ion"]
    else:
        spec = W.ionize_spec(n, r_bubble_max=12.0)
        density = W.density_field_numpy(n, seed=12)
        n_ion = W.nion_from_density(density, fbar=0.7)
        kw = {}
    base = oracle.ionize_grids(spec, density, n_ion, **kw)
    spec.use_mini_halos = 1
    spec.ion_eff_factor_mini = 1.0
    spec.f_limit_mcg = 0.08
    ref = oracle.ionize_grids(spec, density, n_ion, **kw)
    if recomb:
        import torch
        dev = (lambda a: torch.from_numpy(a).cuda()) if device_resident else (lambda a: a)  # noqa: E731
        buf, box, rep = api.ionize_grids(spec, dev(density), dev(n_ion),
                                         **{k: dev(v) for k, v in kw.items()})
        host = (lambda a: a.cpu().numpy()) if device_resident else (lambda a: a)  # noqa: E731
        got = {k: host(getattr(buf, k)) for k in ("neutral_fraction", "z_reion", "kinetic_temperature",
                                                  "ionisation_rate_G12", "mean_free_path")}
        got["report"], mean_m = rep, box.mean_f_coll_MINI
        flags = (got["mean_free_path"] > 0, ref["mean_free_path"] > 0)
        compare(got, ref, spec, flags=flags)
        same = flags[0] == flags[1]
        np.testing.assert_allclose(got["ionisation_rate_G12"][same], ref["ionisation_rate_G12"][same],
                                   rtol=2e-4, atol=1e-7)
    else:
        got = run_plain(api, spec, density, n_ion, device_resident=device_resident)
        compare(got, ref, spec)
        mean_m = None
    n_r = spec.n_radii
    assert list(ref["report"].f_coll_grid_mean_mini[:n_r]) == [0.08] * n_r
    assert list(got["report"].f_coll_grid_mean_mini[:n_r]) == [0.08] * n_r
    assert ref["mean_f_coll_MINI"] == 0.08 and (mean_m is None or mean_m == 0.08)
    assert (ref["neutral_fraction"] == 0).sum() > (base["neutral_fraction"] == 0).sum()


def test_compute_ionized_box_lagrangian_with_mini_halos(gpu_lib, oracle, tmp_path):
    """ComputeIonizedBox, Lagrangian grids + USE_MINI_HALOS: HaloBox.n_ion already holds both
    populations; the call adds the floor of the second one to the barrier, returns it as
    mean_f_coll_MINI, takes the turnover averages from the HaloBox and resets the first
    snapshot's previous box as upstream does."""
    import ctypes as C

    from test_gpu_abi import Session, fptr, ionize_spec_from_scalars
    from test_host_scalars import ScalingConsts
    import test_host_minihalos as HM

    lib = gpu_lib
    n = 32
    ses = Session(lib, tmp_path, HII_DIM=n, SOURCE_MODEL=2, R_BUBBLE_MAX=12.0, USE_MINI_HALOS=True,
                  ALPHA_STAR_MINI=0.5, Z_HEAT_MAX=20.0)
    HM._bind(lib)
    z = 9.0
    shape = (n, n, n)
    density = W.density_field_numpy(n, seed=11)
    n_ion = W.nion_from_density(density, fbar=0.75)
    out = {"neutral_fraction": np.ones(shape, np.float32), "z_reion": np.zeros(shape, np.float32),
           "kinetic_temperature": np.zeros(shape, np.float32)}
    prev_z = np.zeros(shape, np.float32)
    prev_density = np.zeros(shape, np.float32)
    pf = S.PerturbedFieldStruct(density=fptr(density))
    ppf = S.PerturbedFieldStruct(density=fptr(prev_density))
    prev = S.IonizedBoxStruct(z_reion=fptr(prev_z), mean_f_coll=0.4, mean_f_coll_MINI=0.4)
    hb = S.HaloBoxStruct(n_ion=fptr(n_ion), log10_Mcrit_ACG_ave=8.9, log10_Mcrit_MCG_ave=5.4)
    box = S.IonizedBoxStruct(**{k: fptr(v) for k, v in out.items()})
    ts, ics = S.TsBoxStruct(), S.InitialConditionsStruct()
    st = lib.ComputeIonizedBox(z, 0.0, C.byref(pf), C.byref(ppf), C.byref(prev), C.byref(ts),
                               C.byref(hb), C.byref(ics), C.byref(box))
    assert st == 0, lib.c21cm_last_error()
    assert (prev_density == -1.5).all() and prev.mean_f_coll == 0 and prev.mean_f_coll_MINI == 0
    assert box.log10_Mturnover_ave == 8.9 and box.log10_Mturnover_MINI_ave == 5.4
    sc = ScalingConsts()
    assert lib.c21_set_scaling_constants(z, C.byref(sc)) == 0
    lnMmin, lnMmax = math.log(1e5), math.log(1e16)
    f_limit_mcg = lib.c21_Nion_General_MINI(ses.so.Z_HEAT_MAX, lnMmin, lnMmax, 10 ** 5.4, C.byref(sc))
    assert f_limit_mcg > 0 and box.mean_f_coll_MINI == pytest.approx(f_limit_mcg, rel=1e-12)
    spec = ionize_spec_from_scalars(ses, z, lagrangian=True, tables=False, scalars="lib")
    spec.r_lowest = 0  # M_min = 1e5 Msun with mini-halos: every radius is processed
    spec.f_limit_acg = lib.c21_Nion_General(ses.so.Z_HEAT_MAX, lnMmin, lnMmax, 10 ** 8.9, C.byref(sc))
    spec.use_mini_halos, spec.ion_eff_factor_mini, spec.f_limit_mcg = 1, 1.0, f_limit_mcg
    ref = oracle.ionize_grids(spec, density, n_ion)
    ion_g, ion_r = out["neutral_fraction"] == 0, ref["neutral_fraction"] == 0
    assert 0.02 < ion_r.mean() < 0.98 and np.mean(ion_g != ion_r) <= 2e-4
    same = ion_g == ion_r
    np.testing.assert_allclose(out["neutral_fraction"][same], ref["neutral_fraction"][same],
                               rtol=1e-4, atol=5e-6)
    assert box.mean_f_coll == pytest.approx(ref["mean_f_coll"], rel=1e-5)
    del ses
