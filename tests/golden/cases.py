"""The golden cases: inputs and the calls that turn them into outputs.

Shared by make_golden.py (which runs them through the CPU oracle and writes the .npz files
committed next to this file) and by the tests (oracle vs golden on CPU, HIP vs golden on the
MI355X).  Inputs are stored in the fixtures too, so nothing depends on numpy's RNG or FFT
staying bit-stable.  SURVEY.md 8(c) "Fixtures to commit".
"""

import importlib

import numpy as np

S = importlib.import_module("21cmfast_amd.structs")
W = importlib.import_module("21cmfast_amd.workloads")

N_ION = 32      # IonizeBox cases (rocFFT path on the device)
N_ION_NATIVE = 64  # the same at the smallest size of the native split-layout transform
IC_DIM, IC_HII = 32, 16
FILTER_N, FILTER_LEN = 32, 64.0
FILTER_RADII = {0: 5.0, 1: 5.0, 2: 5.0, 3: 5.0, 4: 5.0}
FILTER_PARAM = {0: 0.0, 1: 0.0, 2: 0.0, 3: 8.0, 4: 7.5}


# ---- IonizeBox -----------------------------------------------------------------------------
def ionize_inputs(n=N_ION):
    density = W.density_field_numpy(n, seed=2024, sigma=0.45)
    return {"density": density, "n_ion": W.nion_from_density(density).astype(np.float32)}


def ionize_spec(kind, n=N_ION):
    """kind: 'lagrangian' (two filtered grids, top-hat + exp-MFP) or 'erfc' (CONST-ION-EFF)."""
    mode = W.FCOLL_STARS if kind == "lagrangian" else W.FCOLL_ERFC
    return W.ionize_spec(n, mode=mode, r_bubble_max=14.0)


def ionize_outputs(run, kind, inp):
    """run(spec, density, n_ion_or_None, need_nion) -> dict with neutral_fraction, z_reion,
    kinetic_temperature, report (need_nion: the Eulerian models fill unnormalised_nion)"""
    spec = ionize_spec(kind, inp["density"].shape[0])
    lag = kind == "lagrangian"
    out = run(spec, inp["density"], inp["n_ion"] if lag else None, not lag)
    n = spec.n_radii
    return {"neutral_fraction": out["neutral_fraction"], "z_reion": out["z_reion"],
            "kinetic_temperature": out["kinetic_temperature"],
            "f_coll_grid_mean": np.array(out["report"].f_coll_grid_mean[:n]),
            "global_xH": np.array(out["report"].global_xH)}


# ---- filters (the reference's known-answer geometry, tests/test_filtering.py:52-81) ----------
def filter_input():
    box = np.zeros((FILTER_N,) * 3, np.float32)
    box[FILTER_N // 2, FILTER_N // 2, FILTER_N // 2] = 1.0
    return box


def filter_outputs(filter_grid):
    """filter_grid(box, box_len, type, R, R_param) -> filtered box; the fixture keeps the three
    central lines and the sum (the full boxes would be 0.6 MB of mostly zeros)."""
    box, c = filter_input(), FILTER_N // 2
    out = {}
    for ft, R in FILTER_RADII.items():
        f = filter_grid(box, FILTER_LEN, ft, R, FILTER_PARAM[ft])
        out[f"filter{ft}_lines"] = np.stack([f[:, c, c], f[c, :, c], f[c, c, :]])
        out[f"filter{ft}_sum"] = np.array(f.sum(dtype=np.float64))
    return out


# ---- InitialConditions -> PerturbedField ------------------------------------------------------
def ics_spec(density_is_input=1):
    n_m = 3 * (IC_DIM // 2) ** 2 + 1
    box_len = 3.0 * IC_HII
    k = 2 * np.pi / box_len * np.sqrt(np.arange(n_m, dtype=np.float64))
    pk = np.zeros(n_m)
    pk[1:] = 30.0 * k[1:] ** -2.0
    vol = np.float32(np.float32(box_len) * np.float32(box_len))
    vol = np.float32(vol * np.float32(box_len))
    spec = S.IcsSpec(dim=IC_DIM, dim_z=IC_DIM, hii_dim=IC_HII, hii_dim_z=IC_HII, box_len=box_len,
                     box_len_z=box_len, volume=float(vol), perturb_algorithm=2,
                     perturb_on_high_res=0, density_is_input=density_is_input, n_m=n_m,
                     pk_by_m=pk.ctypes.data_as(S.c_double_p), seed=1234)
    spec._pk = pk
    return spec


def ics_input():
    """hi-res density the IC step starts from (initial_density path: no RNG on either side)."""
    d = W.density_field_numpy(IC_DIM, seed=77, sigma=1.2).astype(np.float32)
    d = np.maximum(d, -50.0)
    return (d - d.mean(dtype=np.float64)).astype(np.float32)


def perturb_spec():
    return S.PerturbSpec(dim=IC_DIM, dim_z=IC_DIM, hii_dim=IC_HII, hii_dim_z=IC_HII,
                         box_len=3.0 * IC_HII, box_len_z=3.0 * IC_HII, perturb_algorithm=2,
                         perturb_on_high_res=0, keep_3d_velocities=1, smooth_evolved_density=0,
                         density_smooth_radius_mpc=0.6, growth_factor=0.12,
                         init_growth_factor=0.0042, dDdt_over_D=2.1e-17)


IC_LOWRES = ("lowres_density", "lowres_vx", "lowres_vy", "lowres_vz", "lowres_vx_2LPT",
             "lowres_vy_2LPT", "lowres_vz_2LPT")


def ics_perturb_outputs(new_ics_arrays, ics_grids, perturb_grids, hires_density):
    spec = ics_spec()
    start = new_ics_arrays(spec)
    start["hires_density"][...] = hires_density
    ic = ics_grids(spec, start)
    pt = perturb_grids(perturb_spec(), ic)
    out = {f"ic_{k}": np.asarray(ic[k]) for k in IC_LOWRES}
    out.update({f"pt_{k}": np.asarray(v) for k, v in pt.items()})
    return out


# ---- spin-temperature filter stage (SpinTemperatureBox.c:560-808) ------------------------------
TS_N, TS_LEN = 24, 48.0
TS_RADII = [1.0, 2.5, 5.0, 9.0]


def tsfilter_inputs():
    rng = np.random.default_rng(606)
    x = np.arange(TS_N)[:, None, None] / TS_N
    dens = (0.3 * rng.standard_normal((TS_N,) * 3) + 0.5 * np.sin(2 * np.pi * x)).astype(np.float32)
    sfr = np.abs(rng.standard_normal((TS_N,) * 3)).astype(np.float32)
    xray = (sfr * sfr).astype(np.float32)
    return {"ts_density": dens, "ts_sfr": sfr, "ts_xray": xray}


def tsfilter_outputs(fill_rbox, annular, inp):
    """fill_rbox(spec, field) -> dict(result, min, average, max); annular(spec, [grids]) ->
    dict(outputs, u_avg, f_avg): top-hat table of the density, one straight-line shell and one
    multiple-scattering shell of (sfr, xray)."""
    out = {}
    r = fill_rbox(S.rbox_spec(TS_N, TS_LEN, TS_RADII, filter_type=0, min_value=-1.0,
                              const_factor=0.2), inp["ts_density"])
    out["rbox_result"] = np.asarray(r["result"])
    out["rbox_stats"] = np.stack([r["min"], r["average"], r["max"]])
    for name, types, r_star in (("sl", [4, 4], 0.0), ("ms", [5, 4], 6.0)):
        a = annular(S.annular_spec(TS_N, TS_LEN, 4.0, 8.0, types, R_star=r_star),
                    [inp["ts_sfr"], inp["ts_xray"]])
        out[f"shell_{name}_sfr"] = np.asarray(a["outputs"][0])
        out[f"shell_{name}_xray"] = np.asarray(a["outputs"][1])
        out[f"shell_{name}_avgs"] = np.stack([a["u_avg"], a["f_avg"]])
    return out


# ---- PerturbedField known-answer geometry (tests/test_perturb.py:52-135 of the reference) ------
def perturb_roll_outputs(perturb_grids):
    """perturb_grids(spec, ics) -> dict(density, velocity_z ...) for the one-cell-displacement
    fake ICs of the reference's test, algorithms 2LPT / Zel'dovich / linear."""
    import sys
    from pathlib import Path

    sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
    import test_oracle_perturb as T

    out = {}
    for algorithm in (2, 1, 0):
        ics = T.fake_ics(algorithm)
        res = perturb_grids(T.perturb_spec(algorithm), ics)
        out[f"density_alg{algorithm}"] = np.asarray(res["density"])
        out[f"expected_alg{algorithm}"] = T.expected_density(ics, algorithm).astype(np.float32)
    return out
