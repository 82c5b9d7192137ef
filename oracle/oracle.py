"""ctypes front-end of the CPU oracle (``oracle/liboracle21.so``).

TEST INFRASTRUCTURE ONLY: imported by ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py``.  Nothing under ``21cmfast_amd/`` imports it.

Parity pinning: see ``oracle/oracle.h`` -- the reference cannot be built or imported in
this image, so the oracle is pinned by the reference's own HDF5 fixtures (same seed, same
universe: ``tests/test_reference_fixtures*.py``) and by its analytic known-answer tests
(restated in ``tests/test_oracle_*.py``).
"""

from __future__ import annotations

import ctypes as C
import importlib
import os
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
LIB_PATH = _HERE / "liboracle21.so"
S = importlib.import_module("21cmfast_amd.structs")

_lib = None


def build(force: bool = False) -> None:
    if force or not LIB_PATH.exists():
        subprocess.run(["make", "-C", str(_HERE)] + (["-B"] if force else []), check=True,
                       stdout=subprocess.DEVNULL)


def load():
    global _lib
    if _lib is None:
        build()
        lib = C.CDLL(str(LIB_PATH))
        vp, i32, f64, f32 = C.c_void_p, C.c_int, C.c_double, C.c_float
        P = C.POINTER
        lib.oracle_fft_r2c.restype = None
        lib.oracle_fft_r2c.argtypes = [vp, i32, i32, i32]
        lib.oracle_fft_c2r.restype = None
        lib.oracle_fft_c2r.argtypes = [vp, i32, i32, i32]
        lib.oracle_filter_window.restype = f64
        lib.oracle_filter_window.argtypes = [i32, f64, f32, f32]
        lib.oracle_filter_grid.restype = i32
        lib.oracle_filter_grid.argtypes = [vp, vp, i32, i32, i32, f64, f64, i32, f64, f64]
        lib.oracle_test_filter.restype = i32
        lib.oracle_test_filter.argtypes = [vp, i32, i32, i32, f64, f64, f64, f64, i32, vp]
        lib.oracle_ionize_grids.restype = i32
        lib.oracle_ionize_grids.argtypes = [
            P(S.IonizeSpec), P(S.PerturbedFieldStruct), P(S.IonizedBoxStruct), P(S.TsBoxStruct),
            P(S.HaloBoxStruct), P(S.IonizedBoxStruct), P(S.IonizeReport),
        ]
        lib.oracle_fully_ionized_temperature.restype = f32
        lib.oracle_fully_ionized_temperature.argtypes = [f32, f32, f32, f32]
        lib.oracle_partially_ionized_temperature.restype = f32
        lib.oracle_partially_ionized_temperature.argtypes = [f32, f32, f32]
        lib.oracle_fgtrm_bias_fast.restype = f64
        lib.oracle_fgtrm_bias_fast.argtypes = [f32, f32, f32, f32, f64]
        lib.oracle_halobox_grids.restype = i32
        lib.oracle_halobox_grids.argtypes = [vp, vp, vp]
        lib.oracle_brightness_grids.restype = i32
        lib.oracle_brightness_grids.argtypes = [vp, vp, vp, vp, vp, vp, vp]
        lib.oracle_fill_Rbox_grids.restype = i32
        lib.oracle_fill_Rbox_grids.argtypes = [vp, vp, vp, vp, vp, vp]
        lib.oracle_annular_filter_grids.restype = i32
        lib.oracle_annular_filter_grids.argtypes = [vp, vp, vp, vp, vp]
        lib.oracle_ts_grids.restype = i32
        lib.oracle_ts_grids.argtypes = [vp] * 7
        lib.oracle_ts_first_grids.restype = i32
        lib.oracle_ts_first_grids.argtypes = [vp, vp, vp]
        for nm in ("oracle_kappa_10", "oracle_kappa_10_elec", "oracle_kappa_10_pH", "oracle_alpha_A"):
            getattr(lib, nm).restype = f64
            getattr(lib, nm).argtypes = [f64]
        lib.oracle_lya_heating_efficiency.restype = f64
        lib.oracle_lya_heating_efficiency.argtypes = [f64, f64, f64, vp]
        for nm, at in (("oracle_ms_mu", [f64]), ("oracle_ms_eta", [f64]),
                       ("oracle_hyper_2F3", [f64, f64, f64])):
            getattr(lib, nm).restype = f64
            getattr(lib, nm).argtypes = at
        lib.oracle_filter_window_ms.restype = f64
        lib.oracle_filter_window_ms.argtypes = [f64, f32, f32, f32]
        lib.oracle_set_threads.restype = None
        lib.oracle_set_threads.argtypes = [i32]
        for name, argt in (
            ("oracle_perturb_grids", [P(S.PerturbSpec), P(S.InitialConditionsStruct),
                                      P(S.PerturbedFieldStruct)]),
            ("oracle_ics_grids", [P(S.IcsSpec), P(S.InitialConditionsStruct)]),
        ):
            if hasattr(lib, name):
                getattr(lib, name).restype = i32
                getattr(lib, name).argtypes = argt
        # small test boxes: a few threads beat 256 (fork/join cost dominates tiny loops)
        lib.oracle_set_threads(min(16, os.cpu_count() or 1))
        _lib = lib
    return _lib


def set_threads(n: int) -> None:
    load().oracle_set_threads(int(n))


def set_fft_threads(n: int) -> None:
    """0: the transforms use the threads of the sweeps; 1: single-threaded transforms (the
    reference's effective behaviour, dft.c:83-85)."""
    lib = load()
    lib.oracle_set_fft_threads.restype = None
    lib.oracle_set_fft_threads.argtypes = [C.c_int]
    lib.oracle_set_fft_threads(int(n))


def _ptr(a: np.ndarray):
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)


def fft_r2c(dense: np.ndarray) -> np.ndarray:
    """Forward transform of a dense float32 box; returns the complex64 half-spectrum."""
    nx, ny, nz = dense.shape
    pad = np.zeros((nx, ny, 2 * (nz // 2 + 1)), np.float32)
    pad[:, :, :nz] = dense
    load().oracle_fft_r2c(_ptr(pad), nx, ny, nz)
    return pad.view(np.complex64).reshape(nx, ny, nz // 2 + 1)


def fft_c2r(spec: np.ndarray, nz: int) -> np.ndarray:
    nx, ny, nzc = spec.shape
    pad = np.ascontiguousarray(spec.astype(np.complex64)).view(np.float32).reshape(nx, ny, 2 * nzc)
    pad = pad.copy()
    load().oracle_fft_c2r(_ptr(pad), nx, ny, nz)
    return pad[:, :, :nz].copy()


def filter_grid(box: np.ndarray, box_len: float, filter_type: int, R: float, R_param: float = 0.0,
                box_len_z: float | None = None) -> np.ndarray:
    box = np.ascontiguousarray(box, np.float32)
    nx, ny, nz = box.shape
    out = np.empty_like(box)
    st = load().oracle_filter_grid(_ptr(box), _ptr(out), nx, ny, nz, box_len,
                                   box_len if box_len_z is None else box_len_z, filter_type, R,
                                   R_param)
    if st:
        raise RuntimeError(f"oracle_filter_grid status {st}")
    return out


def window(filter_type: int, k: float, R: float, R_param: float = 0.0) -> float:
    return load().oracle_filter_window(filter_type, k, R, R_param)


def fptr(a):
    return None if a is None else a.ctypes.data_as(S.c_float_p)


def ionize_grids(spec, density, n_ion=None, xe=None, Tneutral=None, prev_z_reion=None,
                 need_nion=False, prev_nrec=None, whalo_sfr=None, mini=None):
    """Run the oracle's ComputeIonizedBox grid algorithm on numpy inputs.

    ``mini`` (USE_MINI_HALOS): dict of numpy arrays prev_density, log10_mturn_acg, log10_mturn_mcg
    [N] and prev_nion, prev_nion_mini [n_radii, N] (the previous box's unnormalised_nion arrays).
    Returns a dict of output arrays plus the report struct.
    """
    shape = density.shape
    out = {
        "neutral_fraction": np.ones(shape, np.float32),
        "z_reion": np.zeros(shape, np.float32),
        "kinetic_temperature": np.zeros(shape, np.float32),
    }
    if need_nion:
        out["unnormalised_nion"] = np.zeros(shape, np.float32)
    if mini is not None:  # wrapper/outputs.py:1538-1543: one grid per filter radius
        out["unnormalised_nion"] = np.zeros((spec.n_radii,) + shape, np.float32)
        out["unnormalised_nion_mini"] = np.zeros((spec.n_radii,) + shape, np.float32)
        spec.prev_density = fptr(mini["prev_density"])
        spec.log10_mturn_acg = fptr(mini["log10_mturn_acg"])
        spec.log10_mturn_mcg = fptr(mini["log10_mturn_mcg"])
    if spec.recomb_model:  # wrapper/outputs.py:1526-1537
        out["ionisation_rate_G12"] = np.zeros(shape, np.float32)
        out["mean_free_path"] = np.zeros(shape, np.float32)
        out["cumulative_recombinations"] = np.zeros(shape if spec.recomb_model == 2 else (1, 1, 1),
                                                    np.float32)
    pf = S.PerturbedFieldStruct(density=fptr(density))
    prev = S.IonizedBoxStruct(z_reion=fptr(prev_z_reion), cumulative_recombinations=fptr(prev_nrec))
    if mini is not None:
        prev.unnormalised_nion = fptr(mini["prev_nion"])
        prev.unnormalised_nion_mini = fptr(mini["prev_nion_mini"])
    ts = S.TsBoxStruct(xray_ionised_fraction=fptr(xe), kinetic_temp_neutral=fptr(Tneutral))
    hb = S.HaloBoxStruct(n_ion=fptr(n_ion), whalo_sfr=fptr(whalo_sfr))
    box = S.IonizedBoxStruct(
        neutral_fraction=fptr(out["neutral_fraction"]), z_reion=fptr(out["z_reion"]),
        kinetic_temperature=fptr(out["kinetic_temperature"]),
        unnormalised_nion=fptr(out.get("unnormalised_nion")),
        unnormalised_nion_mini=fptr(out.get("unnormalised_nion_mini")),
        ionisation_rate_G12=fptr(out.get("ionisation_rate_G12")),
        mean_free_path=fptr(out.get("mean_free_path")),
        cumulative_recombinations=fptr(out.get("cumulative_recombinations")),
    )
    rep = S.IonizeReport()
    st = load().oracle_ionize_grids(C.byref(spec), C.byref(pf), C.byref(prev), C.byref(ts),
                                    C.byref(hb), C.byref(box), C.byref(rep))
    if st:
        raise RuntimeError(f"oracle_ionize_grids status {st}")
    out["mean_f_coll"] = box.mean_f_coll
    out["mean_f_coll_MINI"] = box.mean_f_coll_MINI
    out["report"] = rep
    return out


def mturn_grids(spec, prev_G12, prev_z_reion, J_21_LW, vcb=None):
    """calculate_mcrit_boxes on numpy arrays -> (log10 M_turn,a, log10 M_turn,m, <a>, <m>)."""
    lib = load()
    lib.oracle_mturn_grids.restype = C.c_int
    lib.oracle_mturn_grids.argtypes = [C.POINTER(S.MturnSpec)] + [S.c_float_p] * 6 + [
        C.POINTER(C.c_double)] * 2
    a = np.zeros(J_21_LW.shape, np.float32)
    m = np.zeros(J_21_LW.shape, np.float32)
    ave_a, ave_m = C.c_double(), C.c_double()
    st = lib.oracle_mturn_grids(C.byref(spec), fptr(prev_G12), fptr(prev_z_reion), fptr(J_21_LW),
                                fptr(vcb), fptr(a), fptr(m), C.byref(ave_a), C.byref(ave_m))
    if st:
        raise RuntimeError(f"oracle_mturn_grids status {st}")
    return a, m, ave_a.value, ave_m.value


IC_FIELDS = ("lowres_density", "lowres_vx", "lowres_vy", "lowres_vz", "lowres_vx_2LPT",
             "lowres_vy_2LPT", "lowres_vz_2LPT", "hires_density", "hires_vx", "hires_vy",
             "hires_vz", "hires_vx_2LPT", "hires_vy_2LPT", "hires_vz_2LPT", "lowres_vcb")


def ics_struct(ics: dict):
    """InitialConditions struct over a dict of numpy arrays (missing fields -> NULL)."""
    return S.InitialConditionsStruct(**{k: fptr(ics.get(k)) for k in IC_FIELDS})


def perturb_grids(spec, ics: dict):
    """Oracle ComputePerturbedField grid algorithm; returns dict(density, velocity_*)."""
    lo = (spec.hii_dim, spec.hii_dim, spec.hii_dim_z)
    out = {"density": np.zeros(lo, np.float32), "velocity_z": np.zeros(lo, np.float32)}
    if spec.keep_3d_velocities:
        out["velocity_x"] = np.zeros(lo, np.float32)
        out["velocity_y"] = np.zeros(lo, np.float32)
    pf = S.PerturbedFieldStruct(**{k: fptr(v) for k, v in out.items()})
    st = load().oracle_perturb_grids(C.byref(spec), C.byref(ics_struct(ics)), C.byref(pf))
    if st:
        raise RuntimeError(f"oracle_perturb_grids status {st}")
    return out


def new_ics_arrays(spec, with_hires_vel=False):
    """Zeroed arrays as InitialConditions.new allocates them (wrapper/outputs.py:534-581)."""
    lo = (spec.hii_dim, spec.hii_dim, spec.hii_dim_z)
    hi = (spec.dim, spec.dim, spec.dim_z)
    ics = {"hires_density": np.zeros(hi, np.float32), "lowres_density": np.zeros(lo, np.float32)}
    shape = hi if spec.perturb_on_high_res else lo
    pre = "hires" if spec.perturb_on_high_res else "lowres"
    for ax in "xyz":
        ics[f"{pre}_v{ax}"] = np.zeros(shape, np.float32)
        if spec.perturb_algorithm == 2:
            ics[f"{pre}_v{ax}_2LPT"] = np.zeros(shape, np.float32)
    return ics


def ics_grids(spec, ics: dict | None = None):
    """Oracle ComputeInitialConditions grid algorithm (fills and returns the dict)."""
    if ics is None:
        ics = new_ics_arrays(spec)
    st = load().oracle_ics_grids(C.byref(spec), C.byref(ics_struct(ics)))
    if st:
        raise RuntimeError(f"oracle_ics_grids status {st}")
    return ics


def halobox_grids(spec, ics: dict, with_whalo=False, with_xray=False):
    """Oracle ComputeHaloBox integrated branch; returns dict(n_ion, halo_sfr[, whalo_sfr,
    halo_xray])."""
    lo = (spec.hii_dim, spec.hii_dim, spec.hii_dim_z)
    out = {"n_ion": np.zeros(lo, np.float32), "halo_sfr": np.zeros(lo, np.float32)}
    if with_whalo:
        out["whalo_sfr"] = np.zeros(lo, np.float32)
    if with_xray:
        out["halo_xray"] = np.zeros(lo, np.float32)
    if spec.use_mini_halos:
        out["halo_sfr_mini"] = np.zeros(lo, np.float32)
    hb = S.HaloBoxStruct(**{k: fptr(v) for k, v in out.items()})
    st = load().oracle_halobox_grids(C.byref(spec), C.byref(ics_struct(ics)), C.byref(hb))
    if st:
        raise RuntimeError(f"oracle_halobox_grids status {st}")
    return out


def halobox_turnovers(spec, m_turn, below_z_heat_max, n_threads, prev_G12, prev_z_reion, J_21_LW,
                      vcb=None, shape=None):
    """get_log10_turnovers with upstream's per-thread running maximum of the atomic turnover."""
    lib = load()
    lib.oracle_halobox_turnovers.restype = C.c_int
    lib.oracle_halobox_turnovers.argtypes = [C.POINTER(S.MturnSpec), C.c_double, C.c_int, C.c_int] + [
        S.c_float_p] * 6 + [C.POINTER(C.c_double)]
    shape = shape or J_21_LW.shape
    a, m = np.zeros(shape, np.float32), np.zeros(shape, np.float32)
    ave = (C.c_double * 2)()
    st = lib.oracle_halobox_turnovers(C.byref(spec), float(m_turn), int(below_z_heat_max),
                                      int(n_threads), fptr(prev_G12), fptr(prev_z_reion),
                                      fptr(J_21_LW), fptr(vcb), fptr(a), fptr(m), ave)
    if st:
        raise RuntimeError(f"oracle_halobox_turnovers status {st}")
    return a, m, (ave[0], ave[1])


def brightness_grids(spec, density, neutral_fraction, spin_temperature=None):
    """Oracle ComputeBrightnessTemp sweep; returns dict(brightness_temp[, tau_21], mean)."""
    out = {"brightness_temp": np.zeros(density.shape, np.float32)}
    if spec.use_ts_fluct:
        out["tau_21"] = np.zeros(density.shape, np.float32)
    mean = C.c_double()
    st = load().oracle_brightness_grids(C.byref(spec), fptr(density), fptr(neutral_fraction),
                                        fptr(spin_temperature), fptr(out["brightness_temp"]),
                                        fptr(out.get("tau_21")), C.byref(mean))
    if st:
        raise RuntimeError(f"oracle_brightness_grids status {st}")
    out["mean"] = mean.value
    return out


def gaussian_pair(counter: int, seed: int):
    lib = load()
    lib.oracle_gaussian_pair.restype = None
    lib.oracle_gaussian_pair.argtypes = [C.c_uint64, C.c_uint64, C.POINTER(C.c_double),
                                         C.POINTER(C.c_double)]
    a, b = C.c_double(), C.c_double()
    lib.oracle_gaussian_pair(counter, seed, C.byref(a), C.byref(b))
    return a.value, b.value


def fill_Rbox_grids(spec, field):
    """Oracle prepare_filter_boxes + fill_Rbox_table; dict(result, min, average, max)."""
    n_R = spec.n_R
    result = np.zeros((n_R,) + field.shape, np.float32)
    mn, av, mx = ((C.c_double * n_R)() for _ in range(3))
    st = load().oracle_fill_Rbox_grids(C.byref(spec), fptr(field), fptr(result), mn, av, mx)
    if st:
        raise RuntimeError(f"oracle_fill_Rbox_grids status {st}")
    return {"result": result, "min": np.array(mn[:]), "average": np.array(av[:]),
            "max": np.array(mx[:])}


def annular_filter_grids(spec, inputs):
    """Oracle one_annular_filter for each input grid; dict(outputs, u_avg, f_avg)."""
    n = spec.n_grids
    outputs = [np.zeros(a.shape, np.float32) for a in inputs]
    in_p = (C.c_void_p * n)(*[a.ctypes.data for a in inputs])
    out_p = (C.c_void_p * n)(*[a.ctypes.data for a in outputs])
    u, f = (C.c_double * n)(), (C.c_double * n)()
    st = load().oracle_annular_filter_grids(C.byref(spec), in_p, out_p, u, f)
    if st:
        raise RuntimeError(f"oracle_annular_filter_grids status {st}")
    return {"outputs": outputs, "u_avg": np.array(u[:]), "f_avg": np.array(f[:])}


TS_FIELDS = ("spin_temperature", "kinetic_temp_neutral", "xray_ionised_fraction")


def ts_grids(spec, density, previous: dict, source: dict | None = None, filtered_density=None):
    """Oracle per-cell part of ComputeTsBox; dict of the three output boxes + the report
    (+ J_21_LW with spec.use_mini_halos)."""
    shape = density.shape
    out = {k: np.zeros(shape, np.float32) for k in TS_FIELDS}
    prev = S.TsBoxStruct(**{k: fptr(previous[k]) for k in TS_FIELDS})
    box = S.TsBoxStruct(**{k: fptr(out[k]) for k in TS_FIELDS})
    if spec.use_mini_halos:
        out["J_21_LW"] = np.zeros(shape, np.float32)
        box.J_21_LW = fptr(out["J_21_LW"])
    src = S.XraySourceBoxStruct(**{k: fptr(v) for k, v in (source or {}).items()})
    rep = S.TsReport()
    st = load().oracle_ts_grids(C.byref(spec), fptr(density), C.byref(prev), C.byref(src),
                                fptr(filtered_density), C.byref(box), C.byref(rep))
    if st:
        raise RuntimeError(f"oracle_ts_grids status {st}")
    out["report"] = rep
    return out


def ts_mcrit_grid(spec, m_turn, J_21_LW, vcb=None):
    """log10 of the Lyman-Werner turnover mass per cell (prepare_filter_boxes)."""
    lib = load()
    lib.oracle_ts_mcrit_grid.restype = C.c_int
    lib.oracle_ts_mcrit_grid.argtypes = [C.POINTER(S.MturnSpec), C.c_double, S.c_float_p,
                                         S.c_float_p, S.c_float_p]
    out = np.zeros(J_21_LW.shape, np.float32)
    st = lib.oracle_ts_mcrit_grid(C.byref(spec), float(m_turn), fptr(J_21_LW), fptr(vcb), fptr(out))
    if st:
        raise RuntimeError(f"oracle_ts_mcrit_grid status {st}")
    return out


def ts_first_grids(spec, density):
    """Oracle init_first_Ts."""
    out = {k: np.zeros(density.shape, np.float32) for k in TS_FIELDS}
    box = S.TsBoxStruct(**{k: fptr(out[k]) for k in TS_FIELDS})
    st = load().oracle_ts_first_grids(C.byref(spec), fptr(density), C.byref(box))
    if st:
        raise RuntimeError(f"oracle_ts_first_grids status {st}")
    return out


def filter_window_ms(k, R_inner, R_outer, R_star):
    return load().oracle_filter_window_ms(float(k), R_inner, R_outer, R_star)


def halo_props(consts, cat, dim, cell_length, redshift, below_z_heat_max=0, vcb_flucts=0, lw=None,
               vcb=None, J21=None, z_re=None, G12=None):
    """test_halo_props (HaloBox.c:658-779): [n_halos, 12] floats; cat: dict(masses, coords, star_rng,
    sfr_rng, xray_rng); lw = (A_LW, BETA_LW, A_VCB, BETA_VCB, sigma_vcb, vcb_const, M_TURN)."""
    lib = load()
    fp = C.POINTER(C.c_float)
    lib.oracle_halo_props.restype = C.c_int
    lib.oracle_halo_props.argtypes = [C.c_void_p, C.c_ulonglong] + [fp] * 5 + [
        C.POINTER(C.c_int), C.c_double, C.c_double, C.c_int, C.c_int, C.POINTER(C.c_double)] + [fp] * 5
    arr = lambda a: None if a is None else np.ascontiguousarray(a, np.float32)  # noqa: E731
    ins = [arr(cat[k]) for k in ("masses", "coords", "star_rng", "sfr_rng", "xray_rng")]
    grids = [arr(g) for g in (vcb, J21, z_re, G12)]
    n = ins[0].size
    out = np.zeros((n, 12), np.float32)
    ptr = lambda a: None if a is None else a.ctypes.data_as(fp)  # noqa: E731
    st = lib.oracle_halo_props(C.byref(consts), n, *[ptr(a) for a in ins], (C.c_int * 3)(*dim),
                               float(cell_length), float(redshift), int(below_z_heat_max), int(vcb_flucts),
                               (C.c_double * 7)(*(lw or (0.0,) * 7)), *[ptr(g) for g in grids], ptr(out))
    if st:
        raise RuntimeError(f"oracle_halo_props status {st}")
    return out
