"""GPU: the halo-catalogue branch of ComputeHaloBox (sum_halos_onto_grid, HaloBox.c:518-560;
move_halo_galprops, map_mass.c:346-476) on the MI355X against the oracle, which deposits in
catalogue order into float grids like upstream with N_THREADS = 1 (the device accumulates in double
and narrows once: tolerances of tests/test_gpu_halobox.py), and through the entry point with
SOURCE_MODEL = CHMF-SAMPLER."""
import ctypes as C
import importlib

import numpy as np
import pytest

import halobox_mini_helpers as HM
from halo_catalogue_helpers import attach, halo_consts, random_catalogue
from test_gpu_halobox import api, compare  # noqa: F401  (fixture)
from test_oracle_halobox import halobox_spec, make_tables, random_ics, with_xray

pytestmark = pytest.mark.gpu
S = importlib.import_module("21cmfast_amd.structs")


@pytest.mark.parametrize("n,N,hires,n_halos,skip,device,kw", [
    (16, 32, False, 5000, False, False, {}),
    (32, 64, True, 200000, False, True, {}),
    (24, 72, True, 30000, True, False, {}),
    (40, 40, False, 100000, True, True, dict(scaling_median=1)),
    (33, 66, False, 40000, False, False, dict(upper_stellar_turnover=0, sigma_sfr_lim=0.0)),
])
def test_catalogue_matches_oracle(api, oracle, n, N, hires, n_halos, skip, device, kw):
    tables = make_tables()

    def spec():
        s = with_xray(halobox_spec(n, N, hires, tables), tables)
        return attach(s, cat, halo_consts(**kw), skip_integral=skip)

    cat = random_catalogue(n_halos, 1.5 * n, seed=n + n_halos)
    ics = random_ics(n, N, hires, seed=n + N, vscale=4.0)
    ref = oracle.halobox_grids(spec(), ics, with_whalo=True, with_xray=True)
    if device:
        import torch

        ics = {k: torch.from_numpy(v).cuda() for k, v in ics.items()}
    got = api.halobox_grids(spec(), ics, with_whalo=True, with_xray=True)
    compare(got, ref)
    assert all(ref[k].max() > 0 for k in ref)
    if skip:  # whalo_sfr comes from the halos, not from n_ion / t_h / t_star
        assert not np.allclose(ref["whalo_sfr"], ref["n_ion"] * 0.37, rtol=1e-3)


def test_device_resident_catalogue(api, oracle):
    """Catalogue arrays already in HBM are used in place."""
    import torch

    n = 24
    tables = make_tables()
    cat = random_catalogue(50000, 1.5 * n, seed=3)
    ics = random_ics(n, n, False, seed=3, vscale=2.0)
    ref = oracle.halobox_grids(attach(halobox_spec(n, n, False, tables), cat, halo_consts(use_xray=0)), ics)
    dev = {k: torch.from_numpy(v).cuda() for k, v in cat.items()}
    hc = S.HaloCatalogStruct(n_halos=50000, buffer_size=50000,
                             **{f: C.cast(dev[k].data_ptr(), S.c_float_p) for f, k in (
                                 ("halo_masses", "masses"), ("halo_coords", "coords"), ("star_rng", "star_rng"),
                                 ("sfr_rng", "sfr_rng"), ("xray_rng", "xray_rng"))})
    spec = halobox_spec(n, n, False, tables)
    consts = halo_consts(use_xray=0)
    spec.halos, spec.halo_consts = C.pointer(hc), C.pointer(consts)
    compare(api.halobox_grids(spec, ics), ref)


@pytest.mark.parametrize("skip", [False, True])
def test_catalogue_with_mini_halos(api, oracle, skip):
    """USE_MINI_HALOS: turnover masses CIC-read at the displaced halo, both populations in n_ion,
    the molecularly cooled star formation in halo_sfr_mini."""
    n = 24
    cat = random_catalogue(60000, 1.5 * n, seed=8)
    cat["masses"] = (cat["masses"] * np.where(np.arange(60000) % 2, 1e-3, 1.0)).astype(np.float32)  # 1e5..1e13
    ics = random_ics(n, n, False, seed=n, vscale=5.0)
    consts = dict(use_mini_halos=1)
    ref = oracle.halobox_grids(attach(HM.mini_spec(n), cat, halo_consts(**consts), skip_integral=skip), ics,
                               with_whalo=True, with_xray=True)
    got = api.halobox_grids(attach(HM.mini_spec(n), cat, halo_consts(**consts), skip_integral=skip), ics,
                            with_whalo=True, with_xray=True)
    assert "halo_sfr_mini" in got and ref["halo_sfr_mini"].max() > 0
    compare(got, ref)
    # the molecular population matters: without it n_ion is lower
    no_mini = oracle.halobox_grids(attach(halobox_spec(n, n, False, make_tables()), cat, halo_consts(),
                                          skip_integral=True), ics)
    if skip:
        assert ref["n_ion"].sum(dtype=np.float64) > 1.0001 * no_mini["n_ion"].sum(dtype=np.float64)


def test_missing_catalogue_arrays_are_refused(api):
    n = 16
    spec = halobox_spec(n, n, False, make_tables())
    cat = random_catalogue(100, 1.5 * n, seed=1)
    attach(spec, cat, halo_consts())
    spec.halos.contents.sfr_rng = None
    with pytest.raises(Exception, match="catalogue"):
        api.halobox_grids(spec, random_ics(n, n, False, seed=1))


@pytest.mark.parametrize("sampler_min_mass", [1e8, 1e6])
def test_entry_point_with_a_catalogue(gpu_lib, oracle, tmp_path, sampler_min_mass):
    """ComputeHaloBox, SOURCE_MODEL = CHMF-SAMPLER: the catalogue's halos with the library's own
    scaling constants plus the integral below SAMPLER_MIN_MASS (1e6 < M_min: halos only), against
    the oracle fed with constants and tables restated here."""
    from test_gpu_abi import Session, fptr
    from test_host_scalars import ScalingConsts

    lib = gpu_lib
    n, N = 32, 64
    ses = Session(lib, tmp_path, HII_DIM=n, DIM=N, SOURCE_MODEL=4, SAMPLER_MIN_MASS=sampler_min_mass,
                  USE_TS_FLUCT=True, RECOMB_MODEL=2, M_TURN=10 ** 8.7)
    z = 8.0
    ics = random_ics(n, N, False, seed=9)
    ics["lowres_density"] = (ics["lowres_density"] * 0.5).astype(np.float32)
    cat = random_catalogue(80000, ses.so.BOX_LEN, seed=12)
    cat["masses"] = np.where(cat["masses"] > 0, np.maximum(cat["masses"], sampler_min_mass), 0).astype(np.float32)
    hc = S.halo_catalog(cat["masses"], cat["coords"], cat["star_rng"], cat["sfr_rng"], cat["xray_rng"])
    keys = ("n_ion", "halo_sfr", "halo_xray", "whalo_sfr")
    out = {k: np.zeros((n, n, n), np.float32) for k in keys}
    hb = S.HaloBoxStruct(**{k: fptr(v) for k, v in out.items()})
    icss = S.InitialConditionsStruct(**{k: fptr(v) for k, v in ics.items()})
    lib.ComputeHaloBox.argtypes = [C.c_double] + [C.c_void_p] * 5
    assert lib.ComputeHaloBox(z, C.byref(icss), None, None, None, C.byref(hb)) == 3  # no catalogue
    st = lib.ComputeHaloBox(z, C.byref(icss), C.byref(hc), None, None, C.byref(hb))
    assert st == 0, lib.c21cm_last_error()

    f64 = C.c_double
    lib.c21_set_scaling_constants.restype = C.c_int
    lib.c21_set_scaling_constants.argtypes = [f64, C.POINTER(ScalingConsts)]
    sc = ScalingConsts()
    assert lib.c21_set_scaling_constants(z, C.byref(sc)) == 0
    ap = ses.ap
    consts = halo_consts(
        z, fstar_10=sc.fstar_10, alpha_star=sc.alpha_star, sigma_star=ap.SIGMA_STAR,
        alpha_upper=ap.UPPER_STELLAR_TURNOVER_INDEX, pivot_upper=ap.UPPER_STELLAR_TURNOVER_MASS,
        fstar_7=sc.fstar_7, alpha_star_mini=sc.alpha_star_mini, acg_thresh=sc.acg_thresh,
        baryon_ratio=ses.cp.OMb / ses.cp.OMm, t_h=sc.t_h, t_star=sc.t_star, sigma_sfr_lim=ap.SIGMA_SFR_LIM,
        sigma_sfr_idx=ap.SIGMA_SFR_INDEX, l_x=sc.l_x, l_x_mini=sc.l_x_mini, sigma_xray=ap.SIGMA_LX,
        fesc_10=sc.fesc_10, fesc_7=sc.fesc_7, alpha_esc=sc.alpha_esc, pop2_ion=sc.pop2_ion,
        pop3_ion=sc.pop3_ion, mturn_a_nofb=sc.mturn_a_nofb, mturn_m_nofb=sc.mturn_m_nofb,
        scaling_median=0, upper_stellar_turnover=int(ses.ao.USE_UPPER_STELLAR_TURNOVER), use_xray=1)
    D = lib.dicke(z)
    M_min = lib.c21_minimum_source_mass(z)
    skip = not (M_min < sampler_min_mass)
    assert skip == (sampler_min_mass < 1e7)  # both branches of HaloBox.c:635 are covered
    spec = S.HaloBoxSpec(dim=N, dim_z=N, hii_dim=n, hii_dim_z=n, box_len=ses.so.BOX_LEN,
                         box_len_z=ses.so.BOX_LEN, perturb_on_high_res=0, lpt2=1, growth_factor=D,
                         init_growth_factor=lib.dicke(ses.so.INITIAL_REDSHIFT))
    keep = []
    if not skip:
        lib.c21_Nion_Conditional_table.restype = C.c_int
        lib.c21_Nion_Conditional_table.argtypes = [f64] * 8 + [C.POINTER(ScalingConsts), C.c_int, f64,
                                                               C.POINTER(C.c_float), C.c_int]
        lib.c21_Xray_Conditional_table.restype = C.c_int
        lib.c21_Xray_Conditional_table.argtypes = [f64] * 8 + [C.POINTER(ScalingConsts), C.c_int,
                                                               C.POINTER(C.c_float), C.c_int]
        lib.sigma_z0.restype = f64
        lib.sigma_z0.argtypes = [f64]
        sc_sfrd = ScalingConsts.from_buffer_copy(sc)
        sc_sfrd.fesc_10, sc_sfrd.fesc_7, sc_sfrd.alpha_esc, sc_sfrd.Mlim_Fesc = 1.0, 1.0, 0.0, 0.0
        d = ics["lowres_density"].astype(np.float64) * D
        dmin, dmax = min(0.0, d.min()) * 1.001, max(0.0, d.max()) * 1.001
        M_cell = lib.c21_rhocrit() * ses.cp.OMm * ses.so.BOX_LEN**3 / n**3
        tabs = [(C.c_float * S.NDELTA_TABLE)() for _ in range(3)]
        lims = (D, np.log(M_min), np.log(float(np.float32(sampler_min_mass))), np.log(M_cell),
                lib.sigma_z0(M_cell), dmin, dmax, sc.mturn_a_nofb)
        assert lib.c21_Nion_Conditional_table(*lims, C.byref(sc), 1, -40.0, tabs[0], S.NDELTA_TABLE) == 0
        assert lib.c21_Nion_Conditional_table(*lims, C.byref(sc_sfrd), 1, -50.0, tabs[1], S.NDELTA_TABLE) == 0
        assert lib.c21_Xray_Conditional_table(*lims, C.byref(sc), 1, tabs[2], S.NDELTA_TABLE) == 0
        pre_stars = lib.c21_rhocrit() * ses.cp.OMb * sc.fstar_10
        spec.update(tab_min=dmin, tab_width=(dmax - dmin) / (S.NDELTA_TABLE - 1.0),
                    ln_nion_table=C.cast(tabs[0], S.c_float_p), ln_sfrd_table=C.cast(tabs[1], S.c_float_p),
                    ln_xray_table=C.cast(tabs[2], S.c_float_p),
                    prefactor_nion=pre_stars * sc.fesc_10 * sc.pop2_ion,
                    prefactor_sfr=pre_stars / sc.t_star / sc.t_h, prefactor_wsfr=1 / sc.t_h / sc.t_star,
                    prefactor_xray=lib.c21_rhocrit() * ses.cp.OMm)
        keep.append(tabs)
    attach(spec, cat, consts, skip_integral=skip)
    ref = oracle.halobox_grids(spec, ics, with_whalo=True, with_xray=True)
    compare(out, ref)
    assert all(ref[k].max() > 0 for k in keys)


def _bind_test_halo_props(lib):
    fp = C.POINTER(C.c_float)
    lib.test_halo_props.restype = C.c_int
    lib.test_halo_props.argtypes = [C.c_double] + [fp] * 4 + [C.c_ulonglong] + [fp] * 6
    return lambda a: None if a is None else a.ctypes.data_as(fp)


def _consts_from_library(lib, ses, z, **kw):
    from test_host_scalars import ScalingConsts

    lib.c21_set_scaling_constants.restype = C.c_int
    lib.c21_set_scaling_constants.argtypes = [C.c_double, C.POINTER(ScalingConsts)]
    sc = ScalingConsts()
    assert lib.c21_set_scaling_constants(z, C.byref(sc)) == 0
    ap = ses.ap
    return sc, halo_consts(
        z, fstar_10=sc.fstar_10, alpha_star=sc.alpha_star, sigma_star=ap.SIGMA_STAR,
        alpha_upper=ap.UPPER_STELLAR_TURNOVER_INDEX, pivot_upper=ap.UPPER_STELLAR_TURNOVER_MASS,
        fstar_7=sc.fstar_7, alpha_star_mini=sc.alpha_star_mini, acg_thresh=sc.acg_thresh,
        baryon_ratio=ses.cp.OMb / ses.cp.OMm, t_h=sc.t_h, t_star=sc.t_star, sigma_sfr_lim=ap.SIGMA_SFR_LIM,
        sigma_sfr_idx=ap.SIGMA_SFR_INDEX, l_x=sc.l_x, l_x_mini=sc.l_x_mini, sigma_xray=ap.SIGMA_LX,
        fesc_10=sc.fesc_10, fesc_7=sc.fesc_7, alpha_esc=sc.alpha_esc, pop2_ion=sc.pop2_ion,
        pop3_ion=sc.pop3_ion, mturn_a_nofb=sc.mturn_a_nofb, mturn_m_nofb=sc.mturn_m_nofb,
        scaling_median=int(ses.ao.HALO_SCALING_RELATIONS_MEDIAN),
        upper_stellar_turnover=int(ses.ao.USE_UPPER_STELLAR_TURNOVER), use_xray=int(ses.ao.USE_TS_FLUCT), **kw)


def test_halo_props_entry_point_reference_known_answers(gpu_lib, tmp_path):
    """test_halo_props (the C function behind py21cmfast's convert_halo_properties) on the case of the
    reference's own test_halo_prop_sampling (tests/test_halo_sampler.py:148-237): same masses,
    deviates, parameters and tolerances; H(z) from the library instead of astropy."""
    from test_gpu_abi import Session
    from test_oracle_halobox_catalogue import reference_kat_catalogue, reference_kat_expectations

    lib = gpu_lib
    ses = Session(lib, tmp_path, HII_DIM=16, DIM=32, USE_TS_FLUCT=True, USE_UPPER_STELLAR_TURNOVER=False,
                  M_TURN=1e5, F_STAR10=0.1, ALPHA_STAR=0.0, t_STAR=0.1, L_X=1e40)
    z = 10.0
    masses, rng, cat = reference_kat_catalogue()
    ptr = _bind_test_halo_props(lib)
    f32 = lambda a: np.ascontiguousarray(a, np.float32)  # noqa: E731
    m, xyz, r = f32(masses), f32(cat["coords"]), f32(rng)
    out = np.zeros((m.size, 12), np.float32)
    st = lib.test_halo_props(z, None, None, None, None, m.size, ptr(m), ptr(xyz), ptr(r), ptr(r), ptr(r), ptr(out))
    assert st == 0, lib.c21cm_last_error()
    sc, c = _consts_from_library(lib, ses, z)
    assert c.l_x == pytest.approx(100.0) and c.mturn_a_nofb == 1e5 and c.t_star == pytest.approx(0.1)
    shmr, ssfr, lx = reference_kat_expectations(masses, rng, c, 1.0 / sc.t_h)
    np.testing.assert_allclose(out[:, 1] / out[:, 0], shmr, rtol=1e-4)
    np.testing.assert_allclose(out[:, 2] / out[:, 1], ssfr, rtol=1e-4)
    np.testing.assert_allclose(out[:, 3] / (out[:, 2] * 31556925.9747), lx, rtol=1e-4)
    lib.c21_hubble.restype, lib.c21_hubble.argtypes = C.c_double, [C.c_float]
    assert 1.0 / sc.t_h == pytest.approx(lib.c21_hubble(z), rel=1e-12)


@pytest.mark.parametrize("below,flucts", [(True, True), (True, False), (False, False)])
def test_halo_props_with_feedback_grids_matches_oracle(gpu_lib, oracle, tmp_path, below, flucts):
    """USE_MINI_HALOS: Lyman-Werner / relative-velocity / reionisation feedback of the halo's cell;
    cut halos keep their rows; device-resident arrays give the same numbers."""
    import torch
    from test_gpu_abi import Session

    lib = gpu_lib
    n = 16
    ses = Session(lib, tmp_path, HII_DIM=n, DIM=2 * n, USE_TS_FLUCT=True, USE_MINI_HALOS=True,
                  Z_HEAT_MAX=35.0 if below else 5.0, V_CB_MODEL=2 if flucts else 3, SOURCE_MODEL=2,
                  HALO_SCALING_RELATIONS_MEDIAN=not flucts)
    z = 11.0
    cat = random_catalogue(20000, ses.so.BOX_LEN * 0.9999, seed=21)
    cat["masses"] = (cat["masses"] * np.where(np.arange(20000) % 2, 1e-3, 1.0)).astype(np.float32)
    cat["coords"] = np.abs(cat["coords"]) % np.float32(ses.so.BOX_LEN * 0.9999)  # in the box, as upstream needs
    cat["coords"][7] = [ses.so.BOX_LEN, 0.0, ses.so.BOX_LEN]  # the edge case of HaloBox.c:700-703
    rng = np.random.default_rng(4)
    vcb = (rng.random((n, n, n)) * 40).astype(np.float32)
    j21 = (10 ** rng.uniform(-3, 1, (n, n, n))).astype(np.float32)
    g12 = (10 ** rng.uniform(-2, 0, (n, n, n))).astype(np.float32)
    zre = np.where(rng.random((n, n, n)) < 0.5, rng.uniform(11.5, 16, (n, n, n)), -1.0).astype(np.float32)
    ptr = _bind_test_halo_props(lib)
    out = np.full((20000, 12), -7.0, np.float32)
    arrs = [cat[k] for k in ("masses", "coords", "star_rng", "sfr_rng", "xray_rng")]
    st = lib.test_halo_props(z, ptr(vcb), ptr(j21), ptr(zre), ptr(g12), 20000, *[ptr(a) for a in arrs], ptr(out))
    assert st == 0, lib.c21cm_last_error()
    sc, c = _consts_from_library(lib, ses, z, use_mini_halos=1)
    lw = (ses.ap.A_LW, ses.ap.BETA_LW, ses.ap.A_VCB, ses.ap.BETA_VCB,
          ses.ct.V_CB_AVG * np.sqrt(3 * np.pi / 8), sc.vcb_const, ses.ap.M_TURN)
    ref = oracle.halo_props(c, cat, (n, n, n), ses.so.BOX_LEN / n, z, below, flucts, lw, vcb, j21, zre, g12)
    cut = cat["masses"] == 0
    assert cut.any() and np.all(out[cut] == -7.0)
    # halos without stars (exp(-M_turn / M) underflows): upstream's 0 * exp(r inf - inf) is NaN for
    # r >= 0; the device writes 0 there
    assert np.isnan(ref).any() and np.isfinite(out[~cut]).all()
    np.testing.assert_array_equal(out[np.isnan(ref)], 0.0)
    ok = ~cut
    want = np.nan_to_num(ref, nan=0.0)
    for col in range(12):
        # the metallicity of halos whose stellar mass is below float range is computed from a
        # denormal SFR on the CPU (a few mantissa bits: 1e-4 off); compared where there are stars
        rows = ok & (ref[:, 1] + ref[:, 6] > 0) if col == 11 else ok
        bad = rows & ~np.isclose(out[:, col], want[:, col], rtol=3e-6, atol=1e-36)
        assert not bad.any(), (col, int(bad.sum()), out[bad][:4, col], ref[bad][:4, col], out[bad][:4, [1, 2, 6, 7]],
                               ref[bad][:4, [1, 2, 6, 7]])
    assert (ref[~cut, 6] > 0).any() and (ref[~cut, 10] > 1).any() == below
    # device-resident
    dev = [torch.from_numpy(a).cuda() for a in arrs + [vcb, j21, zre, g12]]
    out_d = torch.zeros((20000, 12), dtype=torch.float32, device="cuda")
    fp = C.POINTER(C.c_float)
    dp = lambda t: C.cast(t.data_ptr(), fp)  # noqa: E731
    assert lib.test_halo_props(z, dp(dev[5]), dp(dev[6]), dp(dev[7]), dp(dev[8]), 20000,
                               *[dp(t) for t in dev[:5]], dp(out_d)) == 0
    torch.cuda.synchronize()
    np.testing.assert_array_equal(out_d.cpu().numpy()[~cut], out[~cut])
    del ok


def test_grid_totals_equal_the_sum_over_halos_at_scale(gpu_lib, tmp_path):
    """Size-independent property at 256^3 with 5e6 halos: the CIC weights of a halo sum to one, so
    every HaloBox grid times the cell volume integrates to the sum of that property over the
    catalogue -- the grids from ComputeHaloBox (halos only: SAMPLER_MIN_MASS below M_min), the
    per-halo values from test_halo_props; two entry points, two kernels."""
    import torch
    from test_gpu_abi import Session, fptr

    lib = gpu_lib
    n, nh = 256, 5_000_000
    ses = Session(lib, tmp_path, HII_DIM=n, DIM=n, BOX_LEN=384.0, SOURCE_MODEL=4, SAMPLER_MIN_MASS=1e6,
                  USE_TS_FLUCT=True, RECOMB_MODEL=2, PERTURB_ON_HIGH_RES=False)
    z = 9.0
    cat = random_catalogue(nh, 384.0, seed=31)
    dev = {k: torch.from_numpy(v).cuda() for k, v in cat.items()}
    g = torch.Generator(device="cuda").manual_seed(5)
    ics = {k: torch.randn((n, n, n), device="cuda", generator=g) * s for k, s in (
        ("lowres_density", 1.0), ("lowres_vx", 3.0), ("lowres_vy", 3.0), ("lowres_vz", 3.0),
        ("lowres_vx_2LPT", 1.0), ("lowres_vy_2LPT", 1.0), ("lowres_vz_2LPT", 1.0))}
    fp = C.POINTER(C.c_float)
    dp = lambda t: C.cast(t.data_ptr(), fp)  # noqa: E731
    hc = S.HaloCatalogStruct(n_halos=nh, buffer_size=nh, halo_masses=dp(dev["masses"]),
                             halo_coords=dp(dev["coords"]), star_rng=dp(dev["star_rng"]),
                             sfr_rng=dp(dev["sfr_rng"]), xray_rng=dp(dev["xray_rng"]))
    keys = ("n_ion", "halo_sfr", "halo_xray", "whalo_sfr")
    out = {k: torch.zeros((n, n, n), device="cuda") for k in keys}
    hb = S.HaloBoxStruct(**{k: dp(v) for k, v in out.items()})
    icss = S.InitialConditionsStruct(**{k: dp(v) for k, v in ics.items()})
    lib.ComputeHaloBox.argtypes = [C.c_double] + [C.c_void_p] * 5
    assert lib.ComputeHaloBox(z, C.byref(icss), C.byref(hc), None, None, C.byref(hb)) == 0, lib.c21cm_last_error()
    _bind_test_halo_props(lib)
    props = torch.zeros((nh, 12), device="cuda")
    assert lib.test_halo_props(z, None, None, None, None, nh, dp(dev["masses"]), dp(dev["coords"]),
                               dp(dev["star_rng"]), dp(dev["sfr_rng"]), dp(dev["xray_rng"]), dp(props)) == 0
    torch.cuda.synchronize()
    cell_volume = (384.0 / n) ** 3
    for key, col in (("n_ion", 4), ("halo_sfr", 2), ("halo_xray", 3), ("whalo_sfr", 5)):
        total = float(out[key].double().sum()) * cell_volume
        want = float(props[:, col].double().sum())
        assert want > 0 and total == pytest.approx(want, rel=2e-6), key
        assert float(out[key].min()) >= 0
    del fptr
