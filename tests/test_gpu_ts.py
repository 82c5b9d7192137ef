"""Per-cell part of ComputeTsBox on the MI355X against the CPU oracle (reference:
src/py21cmfast/src/SpinTemperatureBox.c:892-927 init_first_Ts, :1010-1086 SFRD from the filtered
density, :1210-1383 get_Ts_fast, :1499-1848 the R loop and the cell outputs).

Tolerances: everything is evaluated in double from float inputs and rounded to float at the end;
the device's exp / log / pow / cbrt differ from glibc's in the last bits, and the T_s fixed point
stops at a 1e-3 relative step, so two evaluations that differ in the last bit can stop one
iteration apart: T_s is compared at 2e-3, x_e and T_k at 2e-6."""

import importlib

import numpy as np
import pytest

import ts_helpers as H

pytestmark = pytest.mark.gpu
S = importlib.import_module("21cmfast_amd.structs")


@pytest.fixture(scope="module")
def api(gpu_lib):
    return importlib.import_module("21cmfast_amd.grid_api")


def to_device(x, device):
    import torch

    if not device or x is None:
        return x
    if isinstance(x, dict):
        return {k: to_device(v, device) for k, v in x.items()}
    return torch.from_numpy(x).cuda()


def to_host(x):
    return x.cpu().numpy() if hasattr(x, "cpu") else x


def run_both(api, oracle, spec, d, device):
    ref = oracle.ts_grids(spec, d["density"], d["previous"], d["source"], d["filtered_density"])
    got = api.ts_grids(spec, to_device(d["density"], device), to_device(d["previous"], device),
                       to_device(d["source"], device), to_device(d["filtered_density"], device))
    return got, ref


def compare(got, ref, spec):
    for k, tol in (("xray_ionised_fraction", 2e-6), ("kinetic_temp_neutral", 2e-6),
                   ("spin_temperature", 2e-3)):
        np.testing.assert_allclose(to_host(got[k]), ref[k], rtol=tol, atol=1e-30, err_msg=k)
    # most cells stop the T_s iteration at the same step: then they agree like the others
    rel = np.abs(to_host(got["spin_temperature"]) / ref["spin_temperature"] - 1)
    assert np.mean(rel < 5e-6) > 0.99
    for f in ("Ts_ave", "Tk_ave", "x_e_ave", "J_alpha_ave", "xheat_ave", "xion_ave"):
        assert getattr(got["report"], f) == pytest.approx(getattr(ref["report"], f), rel=1e-5), f


@pytest.mark.parametrize("device", [False, True])
@pytest.mark.parametrize("lagrangian,n,n_step,nz", [
    (True, 24, 12, None), (False, 24, 12, None),
    (True, 20, 40, 28),   # the default N_STEP_TS, non-cubic box
    (False, 16, 40, None),
])
def test_cell_sweep_matches_oracle(api, oracle, lagrangian, n, n_step, nz, device):
    spec, d = H.make(n=n, n_step=n_step, lagrangian=lagrangian, hii_dim_z=nz)
    got, ref = run_both(api, oracle, spec, d, device)
    compare(got, ref, spec)
    if not lagrangian:
        np.testing.assert_allclose(np.array(got["report"].ave_sfrd[:n_step]),
                                   np.array(ref["report"].ave_sfrd[:n_step]), rtol=1e-6)
    # the special cells of the workload (ts_helpers.make) went through their branches
    Ts = to_host(got["spin_temperature"])
    assert np.isfinite(Ts).all()
    assert to_host(got["kinetic_temp_neutral"]).flat[13] == np.float32(6e4)  # frozen above MAX_TK


def test_fcoll_table_mode_matches_oracle(api, oracle):
    """C21CM_TS_SRC_FCOLL_TABLES (CONST-ION-EFF): the source is the dfcoll/dz table, the box mean that
    normalises it is the f_coll table's; the densest cell sits exactly on the last knot."""
    spec, d = H.make(n=20, n_step=16, lagrangian=False, fcoll_tables=True)
    for i in range(spec.n_step):  # upstream's table ends at the maximum itself (no 1.001 margin)
        fd = d["filtered_density"][i]
        spec.tab_width[i] = (float(fd.max()) * spec.zpp_growth[i] - spec.tab_min[i]) / (S.NDELTA_TABLE - 1.0)
    for device in (False, True):
        got, ref = run_both(api, oracle, spec, d, device)
        compare(got, ref, spec)
        np.testing.assert_allclose(np.array(got["report"].ave_sfrd[:16]),
                                   np.array(ref["report"].ave_sfrd[:16]), rtol=1e-6)


@pytest.mark.parametrize("flags", [
    dict(lya_heating=False), dict(xray_heating=False), dict(cmb_heating=False),
    dict(no_light=True), dict(lya_heating=False, xray_heating=False, cmb_heating=False),
])
def test_option_switches(api, oracle, flags):
    for lagrangian in (True, False):
        spec, d = H.make(n=16, n_step=8, lagrangian=lagrangian, **flags)
        got, ref = run_both(api, oracle, spec, d, False)
        compare(got, ref, spec)


def test_first_box(api, oracle):
    """init_first_Ts: redshift >= Z_HEAT_MAX."""
    fs = H.first_spec(n=20, hii_dim_z=24)
    dens = H.smooth_field((20, 20, 24), np.random.default_rng(2), 0.06)
    ref = oracle.ts_first_grids(fs, dens)
    for device in (False, True):
        got = api.ts_first_grids(fs, to_device(dens, device))
        for k in ref:
            np.testing.assert_allclose(to_host(got[k]), ref[k], rtol=2e-6, err_msg=k)


def test_requests_are_validated(api):
    BackendError = importlib.import_module("21cmfast_amd").BackendError
    spec, d = H.make(n=8, n_step=4)
    with pytest.raises(BackendError, match="filtered_sfr"):
        api.ts_grids(spec, d["density"], d["previous"], None, None)
    spec.lya_dEC = None
    with pytest.raises(BackendError, match="USE_LYA_HEATING"):
        api.ts_grids(spec, d["density"], d["previous"], d["source"], None)
    spec, d = H.make(n=8, n_step=4, lagrangian=False)
    with pytest.raises(BackendError, match="filtered densities"):
        api.ts_grids(spec, d["density"], d["previous"], None, None)
    spec.n_step = 500
    with pytest.raises(BackendError, match="shells"):
        api.ts_grids(spec, d["density"], d["previous"], None, d["filtered_density"])


def test_nan_inputs_are_reported(api):
    """A non-finite spin temperature is an InfinityorNaNError upstream (:1884-1904)."""
    BackendError = importlib.import_module("21cmfast_amd").BackendError
    spec, d = H.make(n=8, n_step=4)
    d["previous"]["kinetic_temp_neutral"].flat[40] = np.nan
    with pytest.raises(BackendError):
        api.ts_grids(spec, d["density"], d["previous"], d["source"], None)


def test_compute_ts_box_lagrangian_entry_point(gpu_lib, oracle, tmp_path):
    """ComputeTsBox with a Lagrangian source model: the XraySourceBox grids go straight into the
    cell sweep.  The oracle runs on the spec the library's host side prepared (its scalars are
    pinned by tests/test_host_heating.py and by the reference fixtures), with and without
    LYA_MULTIPLE_SCATTERING (which only changes how the source box was filtered)."""
    import ctypes as C
    from pathlib import Path

    from test_gpu_abi import Session
    from test_host_heating import Tables

    lib = gpu_lib
    n, n_step = 24, 40
    data = Path(__file__).parent / "golden" / "reference" / "_data"
    ses = Session(lib, tmp_path, data_dir=data, HII_DIM=n, DIM=2 * n, BOX_LEN=1.5 * n,
                  SOURCE_MODEL=2, USE_TS_FLUCT=True, USE_LYA_HEATING=False, Z_HEAT_MAX=30.0)
    rng = np.random.default_rng(8)
    shape = (n, n, n)
    z, prev_z = 14.0, 14.6
    density = H.smooth_field(shape, rng, 0.3)
    prev = {"xray_ionised_fraction": np.exp(rng.uniform(np.log(1.5e-4), np.log(4e-4), shape)).astype(np.float32),
            "kinetic_temp_neutral": (9.0 * (1 + 0.6 * density)).astype(np.float32),
            "spin_temperature": np.full(shape, 30.0, np.float32)}
    src = {"filtered_sfr": np.empty((n_step,) + shape, np.float32),
           "filtered_xray": np.empty((n_step,) + shape, np.float32)}
    for i in range(n_step):  # Msun / yr / Mpc^3 and 1e38 erg / s / Mpc^3, falling with look-back
        f = np.exp(H.smooth_field(shape, rng, 0.7 / (1 + 0.2 * i)))
        src["filtered_sfr"][i] = 2e-4 * f * np.exp(-0.12 * i)
        src["filtered_xray"][i] = 6e-2 * f * np.exp(-0.12 * i)
    out = {k: np.zeros(shape, np.float32) for k in api_fields()}
    fp = lambda a: a.ctypes.data_as(S.c_float_p)  # noqa: E731
    pf = S.PerturbedFieldStruct(density=fp(density))
    prevs = S.TsBoxStruct(**{k: fp(v) for k, v in prev.items()})
    outs = S.TsBoxStruct(**{k: fp(v) for k, v in out.items()})
    srcs = S.XraySourceBoxStruct(**{k: fp(v) for k, v in src.items()})
    lib.ComputeTsBox.restype = C.c_int
    lib.ComputeTsBox.argtypes = [C.c_float, C.c_float, C.c_float, C.c_short] + [C.c_void_p] * 5
    st = lib.ComputeTsBox(z, prev_z, z, 0, C.byref(pf), C.byref(srcs), C.byref(prevs), None,
                          C.byref(outs))
    assert st == 0, lib.c21cm_last_error()
    assert 0 < outs.Q_HI <= 1
    # the same spec from the library's host side, the cell algorithm from the oracle
    spec, tab = S.TsSpec(), Tables()
    lib.c21_ts_prepare.restype = C.c_int
    lib.c21_ts_prepare.argtypes = [C.c_float, C.c_float, C.c_float, C.c_double, C.c_void_p, C.c_void_p]
    x_e_ave = float(prev["xray_ionised_fraction"].sum(dtype=np.float64) / np.float32(density.size))
    assert lib.c21_ts_prepare(z, prev_z, z, x_e_ave, C.byref(spec), C.byref(tab)) == 0
    assert spec.source_mode == S.TS_SRC_GRIDS and spec.no_light == 0
    assert outs.Q_HI == tab.Q_HI
    ref = oracle.ts_grids(spec, density, prev, src, None)
    compare({**out, "report": ref["report"]}, ref, spec)
    # something happened: X-rays ionised and heated, the Lyman-alpha flux coupled T_s to T_k
    assert ref["report"].xion_ave > 0 and ref["report"].J_alpha_ave > 1e-14
    lib.c21_ts_tables_free(C.byref(tab))
    del ses


def api_fields():
    return ("spin_temperature", "kinetic_temp_neutral", "xray_ionised_fraction")


def test_compute_ts_box_first_snapshot_and_refusals(gpu_lib, oracle, tmp_path):
    """redshift >= Z_HEAT_MAX: init_first_Ts from the RECFAST table; unsupported options say so."""
    import ctypes as C
    from pathlib import Path

    from test_gpu_abi import Session

    lib = gpu_lib
    n = 16
    data = Path(__file__).parent / "golden" / "reference" / "_data"
    ses = Session(lib, tmp_path, data_dir=data, HII_DIM=n, SOURCE_MODEL=1, USE_TS_FLUCT=True,
                  USE_LYA_HEATING=False, Z_HEAT_MAX=35.0)
    shape = (n, n, n)
    density = H.smooth_field(shape, np.random.default_rng(3), 0.05)
    out = {k: np.zeros(shape, np.float32) for k in api_fields()}
    fp = lambda a: a.ctypes.data_as(S.c_float_p)  # noqa: E731
    pf = S.PerturbedFieldStruct(density=fp(density))
    outs = S.TsBoxStruct(**{k: fp(v) for k, v in out.items()})
    lib.ComputeTsBox.restype = C.c_int
    lib.ComputeTsBox.argtypes = [C.c_float, C.c_float, C.c_float, C.c_short] + [C.c_void_p] * 5
    z = 36.0
    assert lib.ComputeTsBox(z, 0.0, z, 0, C.byref(pf), None, None, None, C.byref(outs)) == 0, \
        lib.c21cm_last_error()
    xe, TK = lib.c21_xion_RECFAST(z), lib.c21_T_RECFAST(z)
    assert 1e-4 < xe < 1e-3 and 20 < TK < 40  # RECFAST at z = 36
    np.testing.assert_allclose(out["xray_ionised_fraction"], np.float32(xe), rtol=1e-7)
    cT = float(np.float32(0.58 - 0.006 * (np.float32(z) - 10.0)))
    np.testing.assert_allclose(out["kinetic_temp_neutral"], TK * (1 + cT * density.astype(float)), rtol=3e-6)
    Trad = 2.7255 * (1 + z)
    assert np.all((out["spin_temperature"] > out["kinetic_temp_neutral"].min()) & (out["spin_temperature"] < Trad))
    # below Z_HEAT_MAX the previous box is mandatory
    assert lib.ComputeTsBox(20.0, 20.8, 20.0, 0, C.byref(pf), None, None, None, C.byref(outs)) == 3
    prevs = S.TsBoxStruct(**{k: fp(v) for k, v in out.items()})
    ses.ao.USE_MINI_HALOS = True
    assert lib.ComputeTsBox(20.0, 20.8, 20.0, 0, C.byref(pf), None, C.byref(prevs), None, C.byref(outs)) == 3
    assert b"USE_MINI_HALOS" in lib.c21cm_last_error()
    ses.ao.USE_MINI_HALOS = False
    ses.mo.USE_INTERPOLATION_TABLES = 0
    assert lib.ComputeTsBox(20.0, 20.8, 20.0, 0, C.byref(pf), None, C.byref(prevs), None, C.byref(outs)) == 3
    assert b"USE_INTERPOLATION_TABLES" in lib.c21cm_last_error()
    del ses
