"""Time a full evolution with spin temperatures through the reference's entry points, arrays
resident on the device: ComputeInitialConditions once, then for every node redshift from
Z_HEAT_MAX down to z_end (ZPRIME_STEP_FACTOR apart, as run_coeval / run_lightcone evolve):
ComputePerturbedField -> ComputeTsBox -> ComputeIonizedBox -> ComputeBrightnessTemp
(SOURCE_MODEL = E-INTEGRAL, USE_TS_FLUCT, the reference's data tables).  Diagnostic, GPU box only.

    python tools/time_coeval_ts.py [HII_DIM] [DIM] [z_end] [step] [N_THREADS]
"""
import ctypes as C
import importlib
import json
import pathlib
import sys
import tempfile
import time

import numpy as np
import torch

root = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(root))
sys.path.insert(0, str(root / "tests"))
from test_gpu_abi import Session  # noqa: E402

S = importlib.import_module("21cmfast_amd.structs")
pkg = importlib.import_module("21cmfast_amd")

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2 * n
z_end = float(sys.argv[3]) if len(sys.argv) > 3 else 6.0
step = float(sys.argv[4]) if len(sys.argv) > 4 else 1.02
n_threads = int(sys.argv[5]) if len(sys.argv) > 5 else 16
lib = pkg.load(require_gpu=True)
ses = Session(lib, pathlib.Path(tempfile.mkdtemp()), data_dir=root / "tests/golden/reference/_data",
              HII_DIM=n, DIM=N, BOX_LEN=1.5 * n, SOURCE_MODEL=1, USE_TS_FLUCT=True,
              USE_LYA_HEATING=False, HII_FILTER=0, USE_EXP_FILTER=False, CELL_RECOMB=False,
              R_BUBBLE_MAX=30.0, ZPRIME_STEP_FACTOR=step, N_THREADS=n_threads, Z_HEAT_MAX=35.0)
lib.init_heat.restype = C.c_int
assert lib.init_heat() == 0, lib.c21cm_last_error()
f32p = C.POINTER(C.c_float)
dev = lambda shape, fill=0.0: torch.full(shape, fill, dtype=torch.float32, device="cuda")  # noqa: E731
p = lambda t: C.cast(t.data_ptr(), f32p)  # noqa: E731


def timed(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    st = fn()
    torch.cuda.synchronize()
    assert st == 0, lib.c21cm_last_error()
    return (time.perf_counter() - t0) * 1e3


lo, hi = (n, n, n), (N, N, N)
ic = {"hires_density": dev(hi), "lowres_density": dev(lo)}
for ax in "xyz":
    ic[f"lowres_v{ax}"] = dev(lo)
    ic[f"lowres_v{ax}_2LPT"] = dev(lo)
icss = S.InitialConditionsStruct(**{k: p(v) for k, v in ic.items()})
lib.ComputeInitialConditions.argtypes = [C.c_ulonglong, C.c_void_p]
lib.ComputePerturbedField.argtypes = [C.c_float, C.c_void_p, C.c_void_p]
lib.ComputeBrightnessTemp.argtypes = [C.c_float] + [C.c_void_p] * 4
lib.ComputeTsBox.restype = C.c_int
lib.ComputeTsBox.argtypes = [C.c_float, C.c_float, C.c_float, C.c_short] + [C.c_void_p] * 5
zs = 10 ** np.arange(np.log10(1 + z_end), np.log10((1 + 35.0) * step), np.log10(step)) - 1
zs = [float(np.float32(v)) for v in zs[::-1]]
TS = ("spin_temperature", "kinetic_temp_neutral", "xray_ionised_fraction")
res = {"hii_dim": n, "dim": N, "n_snapshots": len(zs), "z_first": zs[0], "z_last": zs[-1],
       "host_threads": n_threads}
res["ics_ms"] = timed(lambda: lib.ComputeInitialConditions(12345, C.byref(icss)))
tot = {"perturb_ms": 0.0, "ts_ms": 0.0, "ionize_ms": 0.0, "brightness_ms": 0.0}
prev_ts_arr = {k: dev(lo) for k in TS}
prev_ts = S.TsBoxStruct(**{k: p(v) for k, v in prev_ts_arr.items()})
prev_ion_arr = {"neutral_fraction": dev(lo, 1.0), "z_reion": dev(lo)}
prev_ion = S.IonizedBoxStruct(**{k: p(v) for k, v in prev_ion_arr.items()})
prev_z, hb = 0.0, S.HaloBoxStruct()
hist = []
t_all = time.perf_counter()
for z in zs:
    dens, vz = dev(lo), dev(lo)
    pf = S.PerturbedFieldStruct(density=p(dens), velocity_z=p(vz))
    tot["perturb_ms"] += timed(lambda: lib.ComputePerturbedField(z, C.byref(icss), C.byref(pf)))
    ts_arr = {k: dev(lo) for k in TS}
    ts = S.TsBoxStruct(**{k: p(v) for k, v in ts_arr.items()})
    tot["ts_ms"] += timed(lambda: lib.ComputeTsBox(z, prev_z, z, 0, C.byref(pf), None, C.byref(prev_ts),
                                                  C.byref(icss), C.byref(ts)))
    ion_arr = {"neutral_fraction": dev(lo, 1.0), "z_reion": dev(lo), "kinetic_temperature": dev(lo),
               "unnormalised_nion": dev(lo)}
    ion = S.IonizedBoxStruct(**{k: p(v) for k, v in ion_arr.items()})
    tot["ionize_ms"] += timed(lambda: lib.ComputeIonizedBox(z, prev_z, C.byref(pf), C.byref(pf),
                                                            C.byref(prev_ion), C.byref(ts), C.byref(hb),
                                                            C.byref(icss), C.byref(ion)))
    bt, tau = dev(lo), dev(lo)
    btb = S.BrightnessTempStruct(brightness_temp=p(bt), tau_21=p(tau))
    tot["brightness_ms"] += timed(lambda: lib.ComputeBrightnessTemp(z, C.byref(ts), C.byref(ion), C.byref(pf),
                                                                    C.byref(btb)))
    hist.append((round(z, 3), round(float(bt.mean()), 3), round(float(ion_arr["neutral_fraction"].mean()), 4),
                 round(float(ts_arr["spin_temperature"].mean()), 2)))
    prev_ts_arr, prev_ts, prev_ion_arr, prev_ion, prev_z = ts_arr, ts, ion_arr, ion, z
res["evolution_s"] = round(time.perf_counter() - t_all, 3)
res.update({k: round(v, 1) for k, v in tot.items()})
res["per_snapshot_ms"] = round(1e3 * res["evolution_s"] / len(zs), 2)
res["history_z_Tb_xH_Ts"] = hist[:: max(1, len(hist) // 12)] + [hist[-1]]
print(json.dumps(res))
