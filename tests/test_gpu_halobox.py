"""GPU: ComputeHaloBox's integrated branch on the MI355X vs the CPU oracle (the oracle adds
floats atomically like the reference, the device accumulates in double and narrows once:
atol 3e-6 * max|field|), the grid extrema helper, and the L-INTEGRAL chain
ComputeHaloBox -> ComputeIonizedBox through the reference's entry points."""

import ctypes as C
import importlib

import numpy as np
import pytest

from test_oracle_halobox import halobox_spec, make_tables, random_ics, with_xray

pytestmark = pytest.mark.gpu
S = importlib.import_module("21cmfast_amd.structs")


@pytest.fixture(scope="module")
def api(gpu_lib):
    return importlib.import_module("21cmfast_amd.grid_api")


def compare(got, ref):
    for k in ref:
        g = got[k].cpu().numpy() if hasattr(got[k], "cpu") else got[k]
        scale = np.abs(ref[k]).max()
        np.testing.assert_allclose(g, ref[k], rtol=2e-5, atol=3e-6 * scale, err_msg=k)


@pytest.mark.parametrize("n,N,hires,vscale,device", [(16, 32, False, 1.0, False),
                                                     (32, 64, True, 1.0, True),
                                                     (24, 72, True, 8.0, False),
                                                     (40, 40, False, 25.0, True),
                                                     (33, 66, True, 3.0, False)])
def test_halobox_matches_oracle(api, oracle, n, N, hires, vscale, device):
    tables = make_tables()
    spec = halobox_spec(n, N, hires, tables)
    ics = random_ics(n, N, hires, seed=n + N, vscale=vscale)
    ref = oracle.halobox_grids(spec, ics, with_whalo=True)
    if device:
        import torch

        ics = {k: torch.from_numpy(v).cuda() for k, v in ics.items()}
    got = api.halobox_grids(spec, ics, with_whalo=True)
    compare(got, ref)


@pytest.mark.parametrize("n,N,hires,vscale,device", [(16, 32, False, 1.0, False),
                                                     (32, 64, True, 6.0, True),
                                                     (40, 40, False, 25.0, True)])
def test_halobox_with_xray_matches_oracle(api, oracle, n, N, hires, vscale, device):
    """Three values per Lagrangian cell (n_ion, SFR, X-ray emissivity): the NV = 3 deposit."""
    tables = make_tables()
    spec = with_xray(halobox_spec(n, N, hires, tables), tables)
    ics = random_ics(n, N, hires, seed=3 * n + N, vscale=vscale)
    ref = oracle.halobox_grids(spec, ics, with_whalo=True, with_xray=True)
    if device:
        import torch

        ics = {k: torch.from_numpy(v).cuda() for k, v in ics.items()}
    got = api.halobox_grids(spec, ics, with_whalo=True, with_xray=True)
    compare(got, ref)
    assert ref["halo_xray"].max() > 0


def test_halobox_to_xray_source_box_chain(gpu_lib, oracle, tmp_path):
    """USE_TS_FLUCT: ComputeHaloBox fills halo_sfr and halo_xray from the initial conditions and
    UpdateXraySourceBox filters them into one shell of the XraySourceBox -- both through the
    reference's entry points; checked against the oracle's annular filter of the same grids."""
    from test_gpu_abi import Session, fptr

    lib = gpu_lib
    n, N = 64, 128
    ses = Session(lib, tmp_path, HII_DIM=n, DIM=N, SOURCE_MODEL=2, USE_TS_FLUCT=True, N_STEP_TS=3)
    z = 9.0
    ics = random_ics(n, N, False, seed=21)
    ics["lowres_density"] = (ics["lowres_density"] * 0.4).astype(np.float32)
    out = {k: np.zeros((n, n, n), np.float32) for k in ("n_ion", "halo_sfr", "halo_xray")}
    hb = S.HaloBoxStruct(**{k: fptr(v) for k, v in out.items()})
    icss = S.InitialConditionsStruct(**{k: fptr(v) for k, v in ics.items()})
    lib.ComputeHaloBox.argtypes = [C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_void_p]
    assert lib.ComputeHaloBox(z, C.byref(icss), None, None, None, C.byref(hb)) == 0, \
        lib.c21cm_last_error()
    assert out["halo_xray"].min() >= 0 and out["halo_xray"].max() > 0
    # L_X per SFR is bounded by L_X (the metallicity factor is <= 1): xray <= sfr * L_X * s_per_yr,
    # up to the differing deposit weights of neighbouring cells
    ratio = out["halo_xray"].sum(dtype=np.float64) / out["halo_sfr"].sum(dtype=np.float64)
    assert 0 < ratio <= ses.ap.L_X * 1e-38 * 31556925.9747 * 1.0001
    # without halo_xray the call is refused
    hb_bad = S.HaloBoxStruct(n_ion=fptr(out["n_ion"]), halo_sfr=fptr(out["halo_sfr"]))
    assert lib.ComputeHaloBox(z, C.byref(icss), None, None, None, C.byref(hb_bad)) == 3
    # one shell of the source box
    ntot = n**3
    fs, fx = np.zeros(3 * ntot, np.float32), np.zeros(3 * ntot, np.float32)
    mean_sfr = np.zeros(3, np.float64)
    xb = S.XraySourceBoxStruct(filtered_sfr=fptr(fs), filtered_xray=fptr(fx),
                               mean_sfr=mean_sfr.ctypes.data_as(C.POINTER(C.c_double)))
    assert lib.UpdateXraySourceBox(C.byref(hb), 3.0, 6.0, 1, 0.0, C.byref(xb)) == 0, \
        lib.c21cm_last_error()
    want = oracle.annular_filter_grids(S.annular_spec(n, ses.so.BOX_LEN, 3.0, 6.0, [4, 4]),
                                       [out["halo_sfr"], out["halo_xray"]])
    for got, ref in ((fs, want["outputs"][0]), (fx, want["outputs"][1])):
        g = got[ntot:2 * ntot].reshape(n, n, n)
        np.testing.assert_allclose(g, ref, rtol=2e-5, atol=2e-5 * np.abs(ref).max())
    assert mean_sfr[1] == pytest.approx(want["f_avg"][0], rel=1e-5)


def test_grid_minmax(api):
    import torch

    a = np.random.default_rng(3).standard_normal(100003).astype(np.float32)
    assert api.grid_minmax(a) == (float(a.min()), float(a.max()))
    assert api.grid_minmax(torch.from_numpy(a).cuda()) == (float(a.min()), float(a.max()))


@pytest.mark.parametrize("median", [False, True])
def test_l_integral_chain_entry_points(gpu_lib, oracle, tmp_path, median):
    """SOURCE_MODEL = L-INTEGRAL: ComputeHaloBox fills n_ion from the initial conditions, and
    ComputeIonizedBox consumes it (two filtered grids, the benchmark path).  median:
    HALO_SCALING_RELATIONS_MEDIAN raises the normalisations of the sub-grid integrals
    (mimic_scatter_in_consts, scaling_relations.c:170-197; checked in tests/test_host_scalars.py)."""
    from test_gpu_abi import Session, call_ionize, fptr, ionize_spec_from_scalars

    lib = gpu_lib
    n, N = 32, 64
    ses = Session(lib, tmp_path, HII_DIM=n, DIM=N, SOURCE_MODEL=2, R_BUBBLE_MAX=12.0,
                  HALO_SCALING_RELATIONS_MEDIAN=median)
    z = 8.0
    ics = random_ics(n, N, False, seed=9)
    ics["lowres_density"] = (ics["lowres_density"] * 0.5).astype(np.float32)
    out = {k: np.zeros((n, n, n), np.float32) for k in ("n_ion", "halo_sfr")}
    hb = S.HaloBoxStruct(n_ion=fptr(out["n_ion"]), halo_sfr=fptr(out["halo_sfr"]))
    icss = S.InitialConditionsStruct(**{k: fptr(v) for k, v in ics.items()})
    lib.ComputeHaloBox.argtypes = [C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_void_p]
    st = lib.ComputeHaloBox(z, C.byref(icss), None, None, None, C.byref(hb))
    assert st == 0, lib.c21cm_last_error()
    assert hb.log10_Mcrit_ACG_ave == pytest.approx(np.log10(ses.ap.M_TURN))
    assert out["n_ion"].min() >= 0 and out["n_ion"].max() > 0 and out["halo_sfr"].max() > 0
    # the same call restated through the grid API with host-side tables from the library
    from test_host_scalars import ScalingConsts

    f64 = C.c_double
    lib.c21_set_scaling_constants.restype = C.c_int
    lib.c21_set_scaling_constants.argtypes = [f64, C.POINTER(ScalingConsts)]
    lib.c21_Nion_Conditional_table.restype = C.c_int
    lib.c21_Nion_Conditional_table.argtypes = [f64] * 8 + [C.POINTER(ScalingConsts), C.c_int, f64,
                                                           C.POINTER(C.c_float), C.c_int]
    lib.sigma_z0.restype = f64
    lib.sigma_z0.argtypes = [f64]
    sc = ScalingConsts()
    assert lib.c21_set_scaling_constants(z, C.byref(sc)) == 0
    if median:
        lib.c21_scaling_consts_mimic_scatter.restype = C.c_int
        lib.c21_scaling_consts_mimic_scatter.argtypes = [C.POINTER(ScalingConsts)]
        f0 = sc.fstar_10
        assert lib.c21_scaling_consts_mimic_scatter(C.byref(sc)) == 0 and sc.fstar_10 > f0
    sc_sfrd = ScalingConsts.from_buffer_copy(sc)
    sc_sfrd.fesc_10, sc_sfrd.fesc_7, sc_sfrd.alpha_esc, sc_sfrd.Mlim_Fesc = 1.0, 1.0, 0.0, 0.0
    D = lib.dicke(z)
    d = ics["lowres_density"].astype(np.float64) * D
    dmin, dmax = min(0.0, d.min()) * 1.001, max(0.0, d.max()) * 1.001
    M_min = lib.c21_minimum_source_mass(z)
    M_cell = lib.c21_rhocrit() * ses.cp.OMm * ses.so.BOX_LEN**3 / n**3
    tabs = [(C.c_float * S.NDELTA_TABLE)(), (C.c_float * S.NDELTA_TABLE)()]
    for tab, consts, floor in ((tabs[0], sc, -40.0), (tabs[1], sc_sfrd, -50.0)):
        assert lib.c21_Nion_Conditional_table(D, np.log(M_min), np.log(1e16), np.log(M_cell),
                                              lib.sigma_z0(M_cell), dmin, dmax, sc.mturn_a_nofb,
                                              C.byref(consts), 1, floor, tab, S.NDELTA_TABLE) == 0
    pre_stars = lib.c21_rhocrit() * ses.cp.OMb * sc.fstar_10
    spec = S.HaloBoxSpec(dim=N, dim_z=N, hii_dim=n, hii_dim_z=n, box_len=ses.so.BOX_LEN,
                         box_len_z=ses.so.BOX_LEN, perturb_on_high_res=0, lpt2=1, growth_factor=D,
                         init_growth_factor=lib.dicke(ses.so.INITIAL_REDSHIFT), tab_min=dmin,
                         tab_width=(dmax - dmin) / (S.NDELTA_TABLE - 1.0),
                         ln_nion_table=C.cast(tabs[0], S.c_float_p),
                         ln_sfrd_table=C.cast(tabs[1], S.c_float_p),
                         prefactor_nion=pre_stars * sc.fesc_10 * sc.pop2_ion,
                         prefactor_sfr=pre_stars / sc.t_star / sc.t_h, prefactor_wsfr=0.0)
    ref = oracle.halobox_grids(spec, ics)
    compare(out, ref)
    # feed it to ComputeIonizedBox; the oracle runs the same two-grid algorithm on the same n_ion
    density = (0.4 * np.random.default_rng(2).standard_normal((n, n, n))).astype(np.float32)
    got = call_ionize(lib, z, density, n_ion=out["n_ion"])
    assert got["status"] == 0, lib.c21cm_last_error()
    ispec = ionize_spec_from_scalars(ses, z, lagrangian=True, tables=False)
    ispec.f_limit_acg = 0.0
    iref = oracle.ionize_grids(ispec, density, out["n_ion"])
    ion_g, ion_r = got["neutral_fraction"] == 0, iref["neutral_fraction"] == 0
    assert np.mean(ion_g != ion_r) <= 2e-4


def test_l_integral_chain_with_recombinations(gpu_lib, oracle, tmp_path):
    """The reference's default astrophysics on fixed grids: SOURCE_MODEL = L-INTEGRAL with
    INHOMO_RECO.  ComputeHaloBox also fills whalo_sfr = n_ion / t_h / t_star (map_mass.c:340-346),
    ComputeIonizedBox filters it next to N_rec and turns it into Gamma_12 at first crossing
    (IonisationBox.c:1126-1131); the oracle runs on the entry points' own source grids with
    the recombination constants restated here."""
    from test_gpu_abi import Session, fptr, ionize_spec_from_scalars, ref_cosmo
    from test_host_scalars import ScalingConsts

    lib = gpu_lib
    n, N = 32, 64
    ses = Session(lib, tmp_path, HII_DIM=n, DIM=N, SOURCE_MODEL=2, R_BUBBLE_MAX=12.0,
                  RECOMB_MODEL=2, CELL_RECOMB=True, USE_EXP_FILTER=True)
    lib.init_MHR.restype = None
    lib.init_MHR()
    z, prev_redshift = 8.0, 8.4
    ics = random_ics(n, N, False, seed=9)
    ics["lowres_density"] = (ics["lowres_density"] * 0.5).astype(np.float32)
    shape = (n, n, n)
    src = {k: np.zeros(shape, np.float32) for k in ("n_ion", "halo_sfr", "whalo_sfr")}
    hb = S.HaloBoxStruct(**{k: fptr(v) for k, v in src.items()})
    icss = S.InitialConditionsStruct(**{k: fptr(v) for k, v in ics.items()})
    lib.ComputeHaloBox.argtypes = [C.c_double] + [C.c_void_p] * 5
    assert lib.ComputeHaloBox(z, C.byref(icss), None, None, None, C.byref(hb)) == 0, \
        lib.c21cm_last_error()
    sc = ScalingConsts()
    lib.c21_set_scaling_constants.restype = C.c_int
    lib.c21_set_scaling_constants.argtypes = [C.c_double, C.POINTER(ScalingConsts)]
    assert lib.c21_set_scaling_constants(z, C.byref(sc)) == 0
    np.testing.assert_allclose(src["whalo_sfr"], src["n_ion"] / sc.t_h / sc.t_star, rtol=3e-7)
    assert src["whalo_sfr"].max() > 0

    rng = np.random.default_rng(2)
    density = np.maximum(0.4 * rng.standard_normal(shape), -0.99).astype(np.float32)
    names = ("neutral_fraction", "z_reion", "kinetic_temperature", "unnormalised_nion",
             "ionisation_rate_G12", "mean_free_path", "cumulative_recombinations")
    arr = {k: np.zeros(shape, np.float32) for k in names}
    arr["neutral_fraction"][...] = 1.0  # the wrapper's initial value (outputs.py:1475-1545)
    box = S.IonizedBoxStruct(**{k: fptr(v) for k, v in arr.items()})
    # a previous snapshot that has already crossed somewhere and has recombined a little
    prev = {"z_reion": np.where(rng.random(shape) < 0.05, np.float32(prev_redshift),
                                np.float32(-1)).astype(np.float32),
            "cumulative_recombinations": (0.1 * rng.random(shape)).astype(np.float32)}
    prev_in = {k: v.copy() for k, v in prev.items()}
    prevs = S.IonizedBoxStruct(**{k: fptr(v) for k, v in prev.items()})
    pf = S.PerturbedFieldStruct(density=fptr(density))
    ts = S.TsBoxStruct()
    st = lib.ComputeIonizedBox(z, prev_redshift, C.byref(pf), C.byref(pf), C.byref(prevs),
                               C.byref(ts), C.byref(hb), C.byref(icss), C.byref(box))
    assert st == 0, lib.c21cm_last_error()

    spec = ionize_spec_from_scalars(ses, z, lagrangian=True, tables=False)
    c = ref_cosmo()
    spec.f_limit_acg = 0.0
    spec.recomb_model, spec.cell_recomb, spec.first_snapshot = 2, 1, 0
    spec.dz = prev_redshift - z
    spec.fabs_dtdz = abs(c.dtdz(z)) / 1e15
    Y_He, m_p = ses.cp.Y_He, 1.6726219e-24
    Ho = c.h * 3.2407e-18
    rho_cgs = 3 * Ho * Ho / (8 * np.pi * 6.6743e-8)
    n_b0 = rho_cgs * c.ob * (1 - Y_He) / m_p + rho_cgs * c.ob * Y_He / (4 * m_p)
    a_uvb = ses.ap.ALPHA_UVB
    spec.gamma_prefactor = ((1 + z) ** 2 * 3.08567758e24 * 6.3e-18 * a_uvb / (a_uvb + 2.75) * n_b0
                            / 1e-12 / spec.rhocrit_omb)
    y = C.POINTER(C.c_double)()
    cc = C.POINTER(C.c_double)()
    lib.c21_rr_tables.restype = C.c_int
    assert lib.c21_rr_tables(C.byref(y), C.byref(cc)) == 0  # pinned by tests/test_host_scalars.py
    spec.rr_y, spec.rr_c = y, cc
    ref = oracle.ionize_grids(spec, density, src["n_ion"], whalo_sfr=src["whalo_sfr"],
                              prev_z_reion=prev_in["z_reion"],
                              prev_nrec=prev_in["cumulative_recombinations"])
    flag_g, flag_r = arr["mean_free_path"] > 0, ref["mean_free_path"] > 0
    assert np.mean(flag_g != flag_r) <= 2e-4
    same = flag_g == flag_r
    assert 0.02 < flag_r.mean() < 0.98
    for k in ("neutral_fraction", "ionisation_rate_G12", "mean_free_path",
              "cumulative_recombinations", "z_reion"):
        np.testing.assert_allclose(arr[k][same], ref[k][same], rtol=2e-4, atol=5e-6, err_msg=k)
    assert arr["ionisation_rate_G12"].max() > 0 and arr["cumulative_recombinations"].max() > 0
    lib.free_MHR.restype = None
    lib.free_MHR()
