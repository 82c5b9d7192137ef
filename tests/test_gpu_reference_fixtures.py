"""The HIP path against vectors the REFERENCE holds (tests/golden/reference/*.h5).

Same pins as tests/test_reference_fixtures.py, but the fields come from the MI355X:
* through the drop-in entry points exactly as py21cmfast would call them
  (Broadcast_struct_global_all with N_THREADS = 2 / 1, ComputeInitialConditions(12345),
  ComputePerturbedField(z)), i.e. with the LIBRARY's own host scalars (sigma_8-normalised EH
  power spectrum, growth factors) and its own restatement of the reference's random stream;
* through the explicit-scalar grid entry points fed from oracle/ref_scalars.py.
DIM = 150 / HII_DIM = 50 are not powers of two, so this also exercises the rocFFT + padded
pipeline.  Tolerances: the reference's own (`atol 5e-3, rtol 1e-3`,
tests/test_integration_features.py:305-308).
"""

import ctypes as C
import importlib

import numpy as np
import pytest

import refpin as RP

pytestmark = pytest.mark.gpu
S = importlib.import_module("21cmfast_amd.structs")


class Broadcast:
    """Parameter structs of the reference's integration-test default, kept alive while in use
    (the library stores POINTERS, as the reference does: InputParameters.c:11-20)."""

    def __init__(self, lib, n_threads=2, **matter):
        self.so = S.default_simulation_options(HII_DIM=RP.HII_DIM, DIM=RP.DIM, BOX_LEN=RP.BOX_LEN,
                                               N_THREADS=n_threads, ZPRIME_STEP_FACTOR=1.04,
                                               SAMPLER_MIN_MASS=1e9)
        self.mo = S.default_matter_options(SOURCE_MODEL=1, **matter)
        self.cp = S.default_cosmo_params()
        self.ap = S.default_astro_params()
        self.ao = S.default_astro_options(USE_EXP_FILTER=False, CELL_RECOMB=False)
        self.ct = S.default_cosmo_tables()
        lib.Broadcast_struct_global_all(C.byref(self.so), C.byref(self.mo), C.byref(self.cp),
                                        C.byref(self.ap), C.byref(self.ao), C.byref(self.ct))
        lib.init_ps()


def fptr(a):
    return a.ctypes.data_as(S.c_float_p)


def run_abi(lib, api, z, n_threads=2, algorithm=2, hires=False):
    keep = Broadcast(lib, n_threads, PERTURB_ALGORITHM=algorithm, PERTURB_ON_HIGH_RES=hires)
    spec = S.IcsSpec(dim=RP.DIM, dim_z=RP.DIM, hii_dim=RP.HII_DIM, hii_dim_z=RP.HII_DIM,
                     perturb_algorithm=algorithm, perturb_on_high_res=int(hires))
    ics = api.new_ics_arrays(spec)
    st = lib.ComputeInitialConditions(RP.SEED, C.byref(api.ics_struct(ics)))
    assert st == 0, lib.c21cm_last_error()
    dens = np.zeros((RP.HII_DIM,) * 3, np.float32)
    vz = np.zeros((RP.HII_DIM,) * 3, np.float32)
    pf = S.PerturbedFieldStruct(density=fptr(dens), velocity_z=fptr(vz))
    st = lib.ComputePerturbedField(z, C.byref(api.ics_struct(ics)), C.byref(pf))
    assert st == 0, lib.c21cm_last_error()
    del keep
    return ics, dens, vz


@pytest.fixture()
def api(gpu_lib):
    return importlib.import_module("21cmfast_amd.grid_api")


@pytest.mark.parametrize("name", list(RP.PT_CASES))
def test_entry_points_reproduce_reference_perturb_field_data(gpu_lib, api, name, monkeypatch):
    monkeypatch.delenv("C21CM_IC_RNG", raising=False)  # default = the reference's stream
    algorithm, hires = RP.PT_CASES[name]
    _, dens, vz = run_abi(gpu_lib, api, 10.0, 2, algorithm, bool(hires))
    worst = RP.check_perturb_fixture(name, dens, vz)
    assert worst < 2e-6  # density power (observed <= 6e-7); the reference asserts rtol 1e-3


@pytest.mark.parametrize("name,n_threads", [("simple", 2), ("sampler_ts_ir_onethread", 1)])
def test_entry_points_reproduce_reference_coeval_powers(gpu_lib, api, name, n_threads,
                                                        monkeypatch):
    monkeypatch.delenv("C21CM_IC_RNG", raising=False)
    ics, dens, vz = run_abi(gpu_lib, api, 18.0, n_threads)
    worst = RP.check_coeval_fields(name, {
        "lowres_density": ics["lowres_density"], "lowres_vx": ics["lowres_vx"],
        "lowres_vx_2LPT": ics["lowres_vx_2LPT"], "density": dens, "velocity_z": vz})
    velocity = worst.pop("velocity_z")  # the reference's dD/dt is quantised in steps of 3.9e-5 here
    assert max(worst.values()) < 5e-6, worst  # (tests/test_reference_fixtures.py); observed 8e-5
    assert velocity < 3e-4


def test_grid_entry_points_match_oracle_on_the_reference_stream(gpu_lib, api, oracle):
    """c21cm_ics_grids with rng_stream = GSL against the oracle, field by field (same P(k)
    table from oracle/ref_scalars.py on both sides), then the fixture through both."""
    spec = RP.ics_spec(2, 0, 2)
    got = api.ics_grids(spec)
    ref = oracle.ics_grids(RP.ics_spec(2, 0, 2))
    for k in ref:
        scale = np.abs(ref[k]).max()
        np.testing.assert_allclose(got[k], ref[k], atol=3e-5 * scale, rtol=1e-4, err_msg=k)
    pf = api.perturb_grids(RP.perturb_spec(10.0), got)
    RP.check_perturb_fixture("simple", pf["density"], pf["velocity_z"])


def test_device_computed_deviates_equal_the_host_path(gpu_lib, api, monkeypatch):
    """The default reference-compatible stream keeps only its serial half on the host: the accepted raw
    word pairs are staged and ln / sqrt of the polar method run on the device (gsl_stream.c:
    c21_gsl_mode_deviates_device); C21CM_GSL_DEVIATES=host computes the deviates with libm on the host.
    The two differ where the device's ln / sqrt round another way: pinned here at 1e-6 of a field's
    maximum with more than 99 % of the cells identical (the default ICs changed silently when
    the device path became the default); the reference fixtures above hold the default path."""
    for threads in (2, 5):  # 5: all five generator kinds (mt19937, gfsr4, cmrg, mrg, taus2) in one draw
        spec = RP.ics_spec(2, 0, threads)
        dev = {k: np.array(v) for k, v in api.ics_grids(spec).items()}
        monkeypatch.setenv("C21CM_GSL_DEVIATES", "host")
        host = api.ics_grids(RP.ics_spec(2, 0, threads))
        monkeypatch.delenv("C21CM_GSL_DEVIATES")
        for k in host:
            scale = np.abs(host[k]).max()
            assert np.abs(dev[k] - host[k]).max() <= 1e-6 * scale, (threads, k)
            assert np.mean(dev[k] == host[k]) > 0.99, (threads, k, np.mean(dev[k] == host[k]))


def test_philox_option_is_another_realisation(gpu_lib, api, monkeypatch):
    """C21CM_IC_RNG=philox keeps the fast device generator: right P(k), other universe."""
    monkeypatch.setenv("C21CM_IC_RNG", "philox")
    ics, _, _ = run_abi(gpu_lib, api, 18.0, 2)
    p, _ = RP.get_power(ics["lowres_density"], RP.BOX_LEN)
    ref = RP.fixture("power_spectra", "simple")["coeval/power_lowres_density"]
    ratio = p / ref
    assert np.abs(ratio - 1).max() > 0.02          # not the same realisation ...
    assert 0.8 < np.median(ratio[3:]) < 1.25        # ... of the same power spectrum


# ---- the whole coeval chain against the reference's z = 18 fixtures ------------------------------
# simple: E-INTEGRAL, no-mdz: CONST-ION-EFF, fixed_halogrids: L-INTEGRAL (HaloBox -> IonizedBox);
# all with HII_FILTER = real-space top-hat, USE_EXP_FILTER = CELL_RECOMB = False, R_BUBBLE_MAX = 15,
# N_THREADS = 2 (reference: tests/produce_integration_test_data.py:48-63,83-90,168-173).
# fftw_wisdom: the options of `simple` with USE_FFTW_WISDOM (:251) -- accepted and ignored here, and
# upstream's own transform results do not depend on the planner's choice beyond round-off
COEVAL_SOURCE = {"simple": 1, "no-mdz": 0, "fixed_halogrids": 2, "fftw_wisdom": 1}


def run_coeval_abi(lib, api, tmp_path, name):
    from test_gpu_abi import Session

    z = 18.0
    ses = Session(lib, tmp_path, HII_DIM=RP.HII_DIM, DIM=RP.DIM, BOX_LEN=RP.BOX_LEN,
                  N_THREADS=2, ZPRIME_STEP_FACTOR=1.04, SOURCE_MODEL=COEVAL_SOURCE[name],
                  HII_FILTER=0, USE_EXP_FILTER=False, CELL_RECOMB=False, R_BUBBLE_MAX=15.0,
                  USE_UPPER_STELLAR_TURNOVER=False, USE_FFTW_WISDOM=(name == "fftw_wisdom"))
    spec = S.IcsSpec(dim=RP.DIM, dim_z=RP.DIM, hii_dim=RP.HII_DIM, hii_dim_z=RP.HII_DIM,
                     perturb_algorithm=2)
    ics = api.new_ics_arrays(spec)
    icss = api.ics_struct(ics)
    assert lib.ComputeInitialConditions(RP.SEED, C.byref(icss)) == 0, lib.c21cm_last_error()
    shape = (RP.HII_DIM,) * 3
    new = lambda v=0.0: np.full(shape, v, np.float32)  # noqa: E731
    dens, vz = new(), new()
    pf = S.PerturbedFieldStruct(density=fptr(dens), velocity_z=fptr(vz))
    assert lib.ComputePerturbedField(z, C.byref(icss), C.byref(pf)) == 0, lib.c21cm_last_error()
    hb_arrays = {"n_ion": new(), "halo_sfr": new()}
    hb = S.HaloBoxStruct(**{k: fptr(v) for k, v in hb_arrays.items()})
    ts, prev_arr = S.TsBoxStruct(), new()
    prev = S.IonizedBoxStruct(z_reion=fptr(prev_arr))
    if name == "fixed_halogrids":
        lib.ComputeHaloBox.restype = C.c_int
        lib.ComputeHaloBox.argtypes = [C.c_double] + [C.c_void_p] * 5
        st = lib.ComputeHaloBox(z, C.byref(icss), None, None, None, C.byref(hb))
        assert st == 0, lib.c21cm_last_error()
    out = {"neutral_fraction": new(1.0), "z_reion": new(), "kinetic_temperature": new(),
           "unnormalised_nion": new()}
    box = S.IonizedBoxStruct(**{k: fptr(v) for k, v in out.items()})
    st = lib.ComputeIonizedBox(z, 0.0, C.byref(pf), C.byref(pf), C.byref(prev), C.byref(ts),
                               C.byref(hb), C.byref(icss), C.byref(box))
    assert st == 0, lib.c21cm_last_error()
    bt = new()
    btb = S.BrightnessTempStruct(brightness_temp=fptr(bt))
    lib.ComputeBrightnessTemp.argtypes = [C.c_float] + [C.c_void_p] * 4
    st = lib.ComputeBrightnessTemp(z, C.byref(ts), C.byref(box), C.byref(pf), C.byref(btb))
    assert st == 0, lib.c21cm_last_error()
    del ses
    return {"density": dens, "velocity_z": vz, "neutral_fraction": out["neutral_fraction"],
            "z_reion": out["z_reion"], "brightness_temp": bt,
            "lowres_density": ics["lowres_density"]}


@pytest.mark.parametrize("name", list(COEVAL_SOURCE))
def test_entry_points_reproduce_reference_coeval_ionization(gpu_lib, api, tmp_path, name,
                                                            monkeypatch):
    """IC -> PerturbedField -> [HaloBox ->] IonizedBox -> BrightnessTemp through the reference's
    entry points, same seed: the binned power of x_HI, z_reion and dT_b of the reference's own
    run.  At z = 18 only a handful of cells cross the barrier, so power_z_reion is white noise
    whose level counts the ionised cells: it matches to 1e-6 only if exactly the same cells
    ionise.  power_neutral_fraction follows the partial ionisations 1 - f_coll zeta of every
    cell and carries the host quadratures (sigma(M), conditional mass function): inside the 1e-4
    the reference prints its own comparison at (tests/test_integration_features.py:69-81)."""
    monkeypatch.delenv("C21CM_IC_RNG", raising=False)
    got = run_coeval_abi(gpu_lib, api, tmp_path, name)
    worst = RP.check_coeval_fields(name, {k: got[k] for k in
                                          ("lowres_density", "density", "velocity_z")})
    velocity = worst.pop("velocity_z")  # the reference's dD/dt is quantised in steps of 3.9e-5 here
    assert max(worst.values()) < 5e-6, worst  # (tests/test_reference_fixtures.py); observed 8e-5
    assert velocity < 3e-4
    f = RP.fixture("power_spectra", name)
    p_z, _ = RP.get_power(got["z_reion"], RP.BOX_LEN)
    np.testing.assert_allclose(p_z, f["coeval/power_z_reion"], rtol=1e-5, atol=1e-9)
    p_x, _ = RP.get_power(got["neutral_fraction"], RP.BOX_LEN)
    np.testing.assert_allclose(p_x, f["coeval/power_neutral_fraction"], rtol=1e-4)  # observed <= 5e-5
    p_b, _ = RP.get_power(got["brightness_temp"], RP.BOX_LEN)
    np.testing.assert_allclose(p_b, f["coeval/power_brightness_temp"], rtol=1e-5)  # observed <= 2.1e-6
    # the lightcone's last node is this redshift: its global x_HI is the box mean
    assert got["neutral_fraction"].mean() == pytest.approx(
        f["lightcone/global_neutral_fraction"][-1], rel=2e-6)


@pytest.mark.parametrize("name,model,cell", [("inhomo", 2, False), ("homo", 1, True)])
def test_entry_points_reproduce_reference_recombination_chain(gpu_lib, api, tmp_path, name, model,
                                                              cell, monkeypatch):
    """run_coeval's evolution for a recombination model through the reference's entry points:
    18 node redshifts from Z_HEAT_MAX down to 18 (step 1.04), every ComputeIonizedBox receiving
    the previous snapshot's box (z_reion, cumulative_recombinations).  Pins Gamma_12 at first
    crossing, the mean free path bookkeeping and the MHR00 recombination-rate tables (init_MHR)
    against the reference's own run."""
    from test_gpu_abi import Session
    from test_reference_fixtures_ionize import check_recomb_fixture, node_redshifts

    monkeypatch.delenv("C21CM_IC_RNG", raising=False)
    lib = gpu_lib
    ses = Session(lib, tmp_path, HII_DIM=RP.HII_DIM, DIM=RP.DIM, BOX_LEN=RP.BOX_LEN, N_THREADS=2,
                  ZPRIME_STEP_FACTOR=1.04, SOURCE_MODEL=1, HII_FILTER=0, USE_EXP_FILTER=False,
                  CELL_RECOMB=cell, R_BUBBLE_MAX=50.0, RECOMB_MODEL=model,
                  USE_UPPER_STELLAR_TURNOVER=False)
    lib.init_MHR.restype = None
    lib.init_MHR()
    spec = S.IcsSpec(dim=RP.DIM, dim_z=RP.DIM, hii_dim=RP.HII_DIM, hii_dim_z=RP.HII_DIM,
                     perturb_algorithm=2)
    ics = api.new_ics_arrays(spec)
    icss = api.ics_struct(ics)
    assert lib.ComputeInitialConditions(RP.SEED, C.byref(icss)) == 0, lib.c21cm_last_error()
    shape = (RP.HII_DIM,) * 3
    rshape = shape if model == 2 else (1, 1, 1)
    names = ("neutral_fraction", "z_reion", "kinetic_temperature", "unnormalised_nion",
             "ionisation_rate_G12", "mean_free_path", "cumulative_recombinations")

    def new_box():
        arr = {k: np.zeros(rshape if k == "cumulative_recombinations" else shape, np.float32)
               for k in names}
        arr["neutral_fraction"][...] = 1.0
        return arr, S.IonizedBoxStruct(**{k: fptr(v) for k, v in arr.items()})

    prev_arr, prev = new_box()  # the zero-filled "initial" previous box (single_field.py)
    prev_z, ts, hb = 0.0, S.TsBoxStruct(), S.HaloBoxStruct()
    for z in node_redshifts():
        dens, vz = np.zeros(shape, np.float32), np.zeros(shape, np.float32)
        pf = S.PerturbedFieldStruct(density=fptr(dens), velocity_z=fptr(vz))
        assert lib.ComputePerturbedField(z, C.byref(icss), C.byref(pf)) == 0
        arr, box = new_box()
        st = lib.ComputeIonizedBox(z, prev_z, C.byref(pf), C.byref(pf), C.byref(prev), C.byref(ts),
                                   C.byref(hb), C.byref(icss), C.byref(box))
        assert st == 0, lib.c21cm_last_error()
        prev_arr, prev, prev_z = arr, box, z
    del ses
    check_recomb_fixture(name, prev_arr)
    assert prev_arr["neutral_fraction"].mean() == pytest.approx(
        RP.fixture("power_spectra", name)["lightcone/global_neutral_fraction"][-1], rel=2e-6)
    # the mean free path of a crossing cell is one of the filter radii
    mfp = prev_arr["mean_free_path"]
    assert (mfp > 0).sum() == (prev_arr["ionisation_rate_G12"] > 0).sum() >= 1
