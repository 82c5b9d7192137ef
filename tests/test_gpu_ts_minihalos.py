"""Spin-temperature cell algorithm with the molecularly cooled population (USE_MINI_HALOS,
E-INTEGRAL) on the MI355X against the oracle: 2-D SFRD tables over the filtered Lyman-Werner
turnover grids, both populations in the shell loop, the J_21_LW output, the turnover-mass grid.
Tolerances as tests/test_gpu_ts.py; J_21_LW (a sum of float-rounded terms in fp64) at 2e-6.
Reference behaviour: SpinTemperatureBox.c:535-565,1011-1075,1642-1733,1843-1845."""
import importlib
import math

import numpy as np
import pytest

import ts_helpers as H
from test_gpu_ts import api, compare, to_device, to_host  # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu
S = importlib.import_module("21cmfast_amd.structs")


def run_both(api, oracle, spec, d, device):
    ref = oracle.ts_grids(spec, d["density"], d["previous"], d["source"], d["filtered_density"])
    host_mcrit = spec._keep["mcrit"]
    if device:  # device-resident turnover grids and tables
        import torch
        spec._keep["mcrit_dev"] = torch.from_numpy(host_mcrit).cuda()
        spec._keep["tabs_mini_dev"] = torch.from_numpy(spec._keep["tabs_mini"]).cuda()
        import ctypes as C
        spec.filtered_log10_mcrit = C.cast(spec._keep["mcrit_dev"].data_ptr(), S.c_float_p)
        spec.ln_sfrd_tables_mini = C.cast(spec._keep["tabs_mini_dev"].data_ptr(), S.c_float_p)
    got = api.ts_grids(spec, to_device(d["density"], device), to_device(d["previous"], device),
                       None, to_device(d["filtered_density"], device))
    if device:
        import torch
        torch.cuda.synchronize()
        spec.filtered_log10_mcrit = host_mcrit.ctypes.data_as(S.c_float_p)
        spec.ln_sfrd_tables_mini = spec._keep["tabs_mini"].ctypes.data_as(S.c_float_p)
    return got, ref


def compare_mini(got, ref, spec):
    compare(got, ref, spec)
    np.testing.assert_allclose(to_host(got["J_21_LW"]), ref["J_21_LW"], rtol=2e-6, atol=1e-30)
    n = spec.n_step
    np.testing.assert_allclose(np.array(got["report"].ave_sfrd_mini[:n]),
                               np.array(ref["report"].ave_sfrd_mini[:n]), rtol=1e-9)


@pytest.mark.parametrize("device", [False, True])
@pytest.mark.parametrize("n,n_step,nz,lya", [(24, 12, None, True), (20, 40, 36, False),
                                             (17, 9, None, True)])
def test_two_population_parity(api, oracle, device, n, n_step, nz, lya):
    spec, d = H.make(n=n, n_step=n_step, lagrangian=False, hii_dim_z=nz, lya_heating=lya)
    H.add_minis(spec, d)
    got, ref = run_both(api, oracle, spec, d, device)
    compare_mini(got, ref, spec)
    assert to_host(got["J_21_LW"]).mean() > 1e-3


def test_dark_mini_population_matches_the_tuned_one_population_kernel(api, oracle):
    spec0, d0 = H.make(n=24, n_step=12, lagrangian=False)
    base = api.ts_grids(spec0, d0["density"], d0["previous"], None, d0["filtered_density"])
    spec, d = H.make(n=24, n_step=12, lagrangian=False)
    H.add_minis(spec, d, strength=0.0)
    got, ref = run_both(api, oracle, spec, d, False)
    compare_mini(got, ref, spec)
    for k in ("kinetic_temp_neutral", "xray_ionised_fraction"):
        np.testing.assert_allclose(got[k], base[k], rtol=1e-6)


@pytest.mark.parametrize("device", [False, True])
@pytest.mark.parametrize("lw_copies", [False, True])
def test_source_grids_with_mini_halos(api, oracle, device, lw_copies):
    spec, d = H.make(n=20, n_step=12, lagrangian=True, hii_dim_z=28)
    H.add_minis_grids(spec, d, lw_copies=lw_copies)
    ref = oracle.ts_grids(spec, d["density"], d["previous"], d["source"], None)
    got = api.ts_grids(spec, to_device(d["density"], device), to_device(d["previous"], device),
                       to_device(d["source"], device), None)
    compare(got, ref, spec)
    np.testing.assert_allclose(to_host(got["J_21_LW"]), ref["J_21_LW"], rtol=2e-6, atol=1e-30)
    assert ref["J_21_LW"].mean() > 1e-3


def test_no_light(api, oracle):
    spec, d = H.make(n=16, n_step=8, lagrangian=False, no_light=True)
    H.add_minis(spec, d)
    got, ref = run_both(api, oracle, spec, d, False)
    compare(got, ref, spec)
    assert (got["J_21_LW"] == 0).all()


def test_mcrit_grid_parity(api, oracle):
    shape = (12, 12, 20)
    rng = np.random.default_rng(3)
    ms = S.MturnSpec(hii_dim=12, hii_dim_z=20, redshift=15.0, vcb_const=20.0, A_LW=2.0, BETA_LW=0.6,
                     A_VCB=1.0, BETA_VCB=1.8, sigma_vcb=25.86 * math.sqrt(3 * math.pi / 8))
    j21 = (2.0 * rng.random(shape) ** 3).astype(np.float32)
    vcb = (40 * rng.random(shape)).astype(np.float32)
    for v in (vcb, None):
        for m_turn in (1e5, 10 ** 6.3):
            got = api.ts_mcrit_grid(ms, m_turn, j21, v)
            np.testing.assert_allclose(got, oracle.ts_mcrit_grid(ms, m_turn, j21, v), rtol=3e-7)
    import torch
    got = api.ts_mcrit_grid(ms, 1e5, torch.from_numpy(j21).cuda(), torch.from_numpy(vcb).cuda())
    np.testing.assert_allclose(got.cpu().numpy(), oracle.ts_mcrit_grid(ms, 1e5, j21, vcb), rtol=3e-7)


def test_refusals(api):
    spec, d = H.make(n=12, n_step=6, lagrangian=True)
    spec.use_mini_halos = 1  # source grids without filtered_sfr_mini
    with pytest.raises(RuntimeError, match="filtered_sfr_mini"):
        api.ts_grids(spec, d["density"], d["previous"], d["source"], None)


@pytest.mark.parametrize("minimize_memory", [False, True])
def test_compute_ts_box_with_mini_halos(gpu_lib, oracle, tmp_path, minimize_memory):
    """ComputeTsBox, E-INTEGRAL with USE_MINI_HALOS: the Lyman-Werner turnover grid from the previous
    box's J_21_LW, its shell-filtered copies, both populations in the shell loop, J_21_LW out.  The
    oracle's cell algorithm runs on the spec and tables the library's host side prepares (checked
    on their own in tests/test_host_heating_minihalos.py) and on the grids of the device's filter
    loops (parity of those: tests/test_gpu_tsfilter.py) -- the tau_X = 1 roots are bracketed to 2 %
    like upstream's, so the host tables are only reproducible from bit-identical box averages."""
    import ctypes as C
    from pathlib import Path

    from test_gpu_abi import Session
    from test_host_heating import Tables

    lib = gpu_lib
    n = 24
    data = Path(__file__).parent / "golden" / "reference" / "_data"
    ses = Session(lib, tmp_path, data_dir=data, HII_DIM=n, DIM=2 * n, BOX_LEN=2.0 * n, SOURCE_MODEL=1,
                  USE_TS_FLUCT=True, USE_LYA_HEATING=False, Z_HEAT_MAX=30.0, USE_MINI_HALOS=True,
                  ALPHA_STAR_MINI=0.5, F_STAR7_MINI=10 ** -2.2, L_X_MINI=10 ** 40.8, V_CB_MODEL=3,
                  MINIMIZE_MEMORY=minimize_memory)
    rng = np.random.default_rng(8)
    shape = (n, n, n)
    z, prev_z = 16.0, 16.7
    density = H.smooth_field(shape, rng, 0.25)
    prev = {"xray_ionised_fraction": np.exp(rng.uniform(np.log(1.5e-4), np.log(4e-4), shape)).astype(np.float32),
            "kinetic_temp_neutral": (9.0 * (1 + 0.6 * density)).astype(np.float32),
            "spin_temperature": np.full(shape, 30.0, np.float32),
            "J_21_LW": (0.4 * np.exp(H.smooth_field(shape, rng, 0.8))).astype(np.float32)}
    fields = ("spin_temperature", "kinetic_temp_neutral", "xray_ionised_fraction", "J_21_LW")
    out = {k: np.zeros(shape, np.float32) for k in fields}
    fp = lambda a: a.ctypes.data_as(S.c_float_p)  # noqa: E731
    pf = S.PerturbedFieldStruct(density=fp(density))
    prevs = S.TsBoxStruct(**{k: fp(v) for k, v in prev.items()})
    outs = S.TsBoxStruct(**{k: fp(v) for k, v in out.items()})
    lib.ComputeTsBox.restype = C.c_int
    lib.ComputeTsBox.argtypes = [C.c_float, C.c_float, C.c_float, C.c_short] + [C.c_void_p] * 5
    st = lib.ComputeTsBox(z, prev_z, z, 0, C.byref(pf), None, C.byref(prevs), None, C.byref(outs))
    assert st == 0, lib.c21cm_last_error()

    # ---- the oracle on the library's host tables
    f64, f32, i32 = C.c_double, C.c_float, C.c_int
    for name, res, args in (("c21_ts_prepare_shells", i32, [f32, f32, f32, C.c_void_p, C.c_void_p]),
                            ("c21_ts_prepare_tables", i32, [f64, C.c_void_p, C.c_void_p]),
                            ("c21_ts_sfrd_tables", i32, [C.POINTER(f64), C.POINTER(f64), C.c_void_p,
                                                         C.c_void_p]),
                            ("c21_lyman_werner_threshold", f64, [f32, f32, f32])):
        getattr(lib, name).restype = res
        getattr(lib, name).argtypes = args
    spec, tab = S.TsSpec(), Tables()
    assert lib.c21_ts_prepare_shells(z, prev_z, z, C.byref(spec), C.byref(tab)) == 0
    n_step = tab.n_step
    ms = S.MturnSpec(hii_dim=n, hii_dim_z=n, redshift=z, vcb_const=ses.ap.V_CB_AVG_DEBUG,
                     A_LW=ses.ap.A_LW, BETA_LW=ses.ap.BETA_LW, A_VCB=ses.ap.A_VCB,
                     BETA_VCB=ses.ap.BETA_VCB, sigma_vcb=ses.ct.V_CB_AVG * math.sqrt(3 * math.pi / 8))
    api = importlib.import_module("21cmfast_amd.grid_api")
    mcrit = api.ts_mcrit_grid(ms, ses.ap.M_TURN, prev["J_21_LW"])
    np.testing.assert_allclose(mcrit, oracle.ts_mcrit_grid(ms, ses.ap.M_TURN, prev["J_21_LW"]), rtol=3e-7)
    rs = S.RboxSpec(hii_dim=n, hii_dim_z=n, box_len=ses.so.BOX_LEN, box_len_z=ses.so.BOX_LEN,
                    filter_type=ses.ao.HEAT_FILTER, n_R=n_step,
                    cell_radius=0.620350491 * ses.so.BOX_LEN / float(n))
    for i in range(n_step):
        rs.R[i] = tab.R_values[i]
    # (upstream floors the turnover grid at the no-background threshold, or at 0 when it filters one
    # shell at a time under MINIMIZE_MEMORY: SpinTemperatureBox.c:1463-1466,1591-1594)
    rs.min_value = 0.0 if minimize_memory else math.log10(lib.c21_lyman_werner_threshold(z, 0.0, 0.0))
    rs.const_factor = 1.0
    fm = api.fill_Rbox_grids(rs, mcrit)
    fm_o = oracle.fill_Rbox_grids(rs, mcrit)
    np.testing.assert_allclose(fm["result"], fm_o["result"], rtol=2e-6)
    for i in range(n_step):
        tab.ave_log10_mturn[i] = fm["average"][i]
    x_e_ave = float(prev["xray_ionised_fraction"].sum(dtype=np.float64) / np.float32(density.size))
    assert lib.c21_ts_prepare_tables(x_e_ave, C.byref(spec), C.byref(tab)) == 0, lib.c21cm_last_error()
    assert spec.no_light == 0 and outs.Q_HI == pytest.approx(tab.Q_HI, rel=1e-6)
    rs.min_value, rs.const_factor = -1.0, 1.0 / lib.dicke(z)
    fd = api.fill_Rbox_grids(rs, density)
    mn = (f64 * 128)(*fd["min"], *([0.0] * (128 - n_step)))
    mx = (f64 * 128)(*fd["max"], *([0.0] * (128 - n_step)))
    assert lib.c21_ts_sfrd_tables(mn, mx, C.byref(spec), C.byref(tab)) == 0, lib.c21cm_last_error()
    keep = np.ascontiguousarray(fm["result"])
    spec.filtered_log10_mcrit = keep.ctypes.data_as(S.c_float_p)
    ref = oracle.ts_grids(spec, density, prev, None, fd["result"])
    compare({**out, "report": ref["report"]}, ref, spec)
    # J_21_LW: float32 transforms move the filtered inputs by ~1e-6, the tables are steep in delta
    np.testing.assert_allclose(out["J_21_LW"], ref["J_21_LW"], rtol=3e-4)
    assert 1e-4 < out["J_21_LW"].mean() < 1e3
    assert ref["report"].ave_sfrd_mini[0] > 0
    # the molecularly cooled population matters at z = 16: without it the gas is ionised less
    ses.ao.USE_MINI_HALOS = False
    out0 = {k: np.zeros(shape, np.float32) for k in fields[:3]}
    outs0 = S.TsBoxStruct(**{k: fp(v) for k, v in out0.items()})
    prevs0 = S.TsBoxStruct(**{k: fp(prev[k]) for k in fields[:3]})
    assert lib.ComputeTsBox(z, prev_z, z, 0, C.byref(pf), None, C.byref(prevs0), None,
                            C.byref(outs0)) == 0, lib.c21cm_last_error()
    assert out["xray_ionised_fraction"].mean() > out0["xray_ionised_fraction"].mean()
    lib.c21_ts_tables_free(C.byref(tab))
    del ses


@pytest.mark.parametrize("multiple_scattering", [False, True])
def test_compute_ts_box_source_grids_with_mini_halos(gpu_lib, oracle, tmp_path, multiple_scattering):
    """ComputeTsBox with a Lagrangian source model and USE_MINI_HALOS: XraySourceBox supplies the
    molecularly cooled star-formation grids, the shells' mean turnover masses (tau_X) and, under
    LYA_MULTIPLE_SCATTERING, the straight-line copies for the Lyman-Werner sums."""
    import ctypes as C
    from pathlib import Path

    from test_gpu_abi import Session
    from test_host_heating import Tables

    lib = gpu_lib
    n, n_step = 20, 40
    data = Path(__file__).parent / "golden" / "reference" / "_data"
    ses = Session(lib, tmp_path, data_dir=data, HII_DIM=n, DIM=2 * n, BOX_LEN=1.5 * n, SOURCE_MODEL=2,
                  USE_TS_FLUCT=True, USE_LYA_HEATING=False, Z_HEAT_MAX=30.0, USE_MINI_HALOS=True,
                  ALPHA_STAR_MINI=0.5, LYA_MULTIPLE_SCATTERING=multiple_scattering)
    rng = np.random.default_rng(8)
    shape = (n, n, n)
    z, prev_z = 14.0, 14.6
    density = H.smooth_field(shape, rng, 0.3)
    prev = {"xray_ionised_fraction": np.exp(rng.uniform(np.log(1.5e-4), np.log(4e-4), shape)).astype(np.float32),
            "kinetic_temp_neutral": (9.0 * (1 + 0.6 * density)).astype(np.float32),
            "spin_temperature": np.full(shape, 30.0, np.float32)}
    src = {k: np.empty((n_step,) + shape, np.float32)
           for k in ("filtered_sfr", "filtered_xray", "filtered_sfr_mini")}
    for i in range(n_step):
        f = np.exp(H.smooth_field(shape, rng, 0.7 / (1 + 0.2 * i)))
        src["filtered_sfr"][i] = 2e-4 * f * np.exp(-0.12 * i)
        src["filtered_xray"][i] = 6e-2 * f * np.exp(-0.12 * i)
        src["filtered_sfr_mini"][i] = 5e-5 * f ** 0.7 * np.exp(-0.05 * i)
    if multiple_scattering:
        src["filtered_sfr_lw"] = (src["filtered_sfr"] * 0.9).astype(np.float32)
        src["filtered_sfr_mini_lw"] = (src["filtered_sfr_mini"] * 1.1).astype(np.float32)
    mean_mcrit = np.ascontiguousarray(5.8 + 0.01 * np.arange(n_step), np.float64)
    fields = ("spin_temperature", "kinetic_temp_neutral", "xray_ionised_fraction", "J_21_LW")
    out = {k: np.zeros(shape, np.float32) for k in fields}
    fp = lambda a: a.ctypes.data_as(S.c_float_p)  # noqa: E731
    pf = S.PerturbedFieldStruct(density=fp(density))
    prevs = S.TsBoxStruct(**{k: fp(v) for k, v in prev.items()})
    outs = S.TsBoxStruct(**{k: fp(v) for k, v in out.items()})
    srcs = S.XraySourceBoxStruct(**{k: fp(v) for k, v in src.items()})
    srcs.mean_log10_Mcrit_LW = mean_mcrit.ctypes.data_as(C.POINTER(C.c_double))
    lib.ComputeTsBox.restype = C.c_int
    lib.ComputeTsBox.argtypes = [C.c_float, C.c_float, C.c_float, C.c_short] + [C.c_void_p] * 5
    st = lib.ComputeTsBox(z, prev_z, z, 0, C.byref(pf), C.byref(srcs), C.byref(prevs), None,
                          C.byref(outs))
    assert st == 0, lib.c21cm_last_error()
    f64, f32, i32 = C.c_double, C.c_float, C.c_int
    lib.c21_ts_prepare_shells.restype = i32
    lib.c21_ts_prepare_shells.argtypes = [f32, f32, f32, C.c_void_p, C.c_void_p]
    lib.c21_ts_prepare_tables.restype = i32
    lib.c21_ts_prepare_tables.argtypes = [f64, C.c_void_p, C.c_void_p]
    spec, tab = S.TsSpec(), Tables()
    assert lib.c21_ts_prepare_shells(z, prev_z, z, C.byref(spec), C.byref(tab)) == 0
    assert spec.source_mode == S.TS_SRC_GRIDS and spec.use_mini_halos == 1
    for i in range(n_step):
        tab.ave_log10_mturn[i] = mean_mcrit[i]
    x_e_ave = float(prev["xray_ionised_fraction"].sum(dtype=np.float64) / np.float32(density.size))
    assert lib.c21_ts_prepare_tables(x_e_ave, C.byref(spec), C.byref(tab)) == 0, lib.c21cm_last_error()
    assert outs.Q_HI == tab.Q_HI
    ref = oracle.ts_grids(spec, density, prev, src, None)
    compare({**out, "report": ref["report"]}, ref, spec)
    np.testing.assert_allclose(out["J_21_LW"], ref["J_21_LW"], rtol=2e-6)
    assert out["J_21_LW"].min() > 0  # (the synthetic source grids carry no physical normalisation)
    # without the mean turnover masses the call is refused
    srcs.mean_log10_Mcrit_LW = None
    assert lib.ComputeTsBox(z, prev_z, z, 0, C.byref(pf), C.byref(srcs), C.byref(prevs), None,
                            C.byref(outs)) == 3
    lib.c21_ts_tables_free(C.byref(tab))
    del ses
