"""Time ComputeIonizedBox through the reference's entry point with device-resident arrays for
the Eulerian source models (E-INTEGRAL = `simple` template, CONST-ION-EFF = `const-zeta`), i.e.
what a py21cmfast user of those templates gets (diagnostic; GPU box only).

    PYTHONPATH=. python tools/time_abi_ionize.py [HII_DIM] [SOURCE_MODEL 0|1] [z]
"""
import ctypes as C
import importlib
import json
import pathlib
import sys
import tempfile
import time

import torch

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent / "tests"))
from test_gpu_abi import Session  # noqa: E402

S = importlib.import_module("21cmfast_amd.structs")
W = importlib.import_module("21cmfast_amd.workloads")
pkg = importlib.import_module("21cmfast_amd")

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
src = int(sys.argv[2]) if len(sys.argv) > 2 else 1
z = float(sys.argv[3]) if len(sys.argv) > 3 else 9.0
use_ts = int(sys.argv[4]) if len(sys.argv) > 4 else 0  # 1: an x_e / T_k box of a spin-temperature run
hii_filter = int(sys.argv[5]) if len(sys.argv) > 5 else 1
lib = pkg.load(require_gpu=True)
tmp = pathlib.Path(tempfile.mkdtemp())
ses = Session(lib, tmp, HII_DIM=n, SOURCE_MODEL=src, HII_FILTER=hii_filter, USE_EXP_FILTER=False,
              CELL_RECOMB=False, R_BUBBLE_MAX=40.0, USE_TS_FLUCT=bool(use_ts))

density = W.density_field_torch(n, seed=5, sigma=0.6)
shape = density.shape
f32p = C.POINTER(C.c_float)


def p(t):
    return C.cast(t.data_ptr(), f32p)


def run():
    xH = torch.ones(shape, device="cuda")
    zre = torch.zeros(shape, device="cuda")
    tk = torch.zeros(shape, device="cuda")
    nion = torch.zeros(shape, device="cuda")
    prev = torch.zeros(shape, device="cuda")
    pf = S.PerturbedFieldStruct(density=p(density))
    prevb = S.IonizedBoxStruct(z_reion=p(prev))
    box = S.IonizedBoxStruct(neutral_fraction=p(xH), z_reion=p(zre), kinetic_temperature=p(tk),
                             unnormalised_nion=p(nion))
    ts, hb, ics = S.TsBoxStruct(), S.HaloBoxStruct(), S.InitialConditionsStruct()
    if use_ts:
        g = torch.Generator(device="cuda").manual_seed(3)
        xe = 0.02 + 0.03 * torch.rand(shape, device="cuda", generator=g)
        tn = 8.0 + 4.0 * torch.rand(shape, device="cuda", generator=g)
        ts = S.TsBoxStruct(xray_ionised_fraction=p(xe), kinetic_temp_neutral=p(tn))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    st = lib.ComputeIonizedBox(z, 0.0, C.byref(pf), C.byref(pf), C.byref(prevb), C.byref(ts),
                               C.byref(hb), C.byref(ics), C.byref(box))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert st == 0, lib.c21cm_last_error()
    tm = (C.c_double * 6)()
    lib.c21cm_last_ionize_timing(tm)
    return dt, float(xH.mean()), list(tm)


run()
times = [run() for _ in range(3)]
best = min(times, key=lambda r: r[0])
print(json.dumps({"hii_dim": n, "source_model": src, "z": z, "use_ts_fluct": use_ts, "hii_filter": hii_filter, "ms": best[0] * 1e3,
                  "global_xH": times[0][1], "host_ms": round(best[2][0], 3), "device_preloop_ms": round(best[2][1], 3),
                  "device_rloop_ms": round(best[2][2], 3), "device_postloop_ms": round(best[2][3], 3),
                  "call_wall_ms": round(best[2][4], 3), "n_radii": int(best[2][5])}))
