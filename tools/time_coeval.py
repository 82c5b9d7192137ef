"""Time a coeval snapshot chain through the reference's entry points with device-resident arrays:
ComputeInitialConditions once, then per redshift ComputePerturbedField -> ComputeHaloBox ->
ComputeIonizedBox -> ComputeBrightnessTemp (SOURCE_MODEL = L-INTEGRAL).  Diagnostic, GPU box only.

    PYTHONPATH=. python tools/time_coeval.py [HII_DIM] [DIM]
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
pkg = importlib.import_module("21cmfast_amd")

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2 * n
lib = pkg.load(require_gpu=True)
ses = Session(lib, pathlib.Path(tempfile.mkdtemp()), HII_DIM=n, DIM=N, BOX_LEN=1.5 * n,
              SOURCE_MODEL=2, R_BUBBLE_MAX=30.0)
f32p = C.POINTER(C.c_float)


def dev(shape, fill=0.0):
    return torch.full(shape, fill, dtype=torch.float32, device="cuda")


def p(t):
    return C.cast(t.data_ptr(), f32p)


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
lib.ComputeHaloBox.argtypes = [C.c_double] + [C.c_void_p] * 5
lib.ComputeBrightnessTemp.argtypes = [C.c_float] + [C.c_void_p] * 4
res = {"hii_dim": n, "dim": N, "ics_ms": None, "snapshots": []}
import os


def fresh_ics(stream):
    """Sampling run (hires_density zeroed first: a non-zero one means "use it as the field")."""
    os.environ["C21CM_IC_RNG"] = stream
    ic["hires_density"].zero_()
    return timed(lambda: lib.ComputeInitialConditions(12345, C.byref(icss)))


fresh_ics("philox")
res["ics_ms"] = fresh_ics("philox")            # counter-based device generator
res["ics_ms_reference_stream"] = fresh_ics("gsl")  # upstream's serial host stream (the default)
for z in (12.0, 10.0, 8.0, 7.0):
    for rep in range(2):
        dens, vz = dev(lo), dev(lo)
        pf = S.PerturbedFieldStruct(density=p(dens), velocity_z=p(vz))
        nion, sfr = dev(lo), dev(lo)
        hb = S.HaloBoxStruct(n_ion=p(nion), halo_sfr=p(sfr))
        xH, zre, tk, prev = dev(lo, 1.0), dev(lo), dev(lo), dev(lo)
        prevb = S.IonizedBoxStruct(z_reion=p(prev))
        box = S.IonizedBoxStruct(neutral_fraction=p(xH), z_reion=p(zre), kinetic_temperature=p(tk))
        bt = dev(lo)
        btb = S.BrightnessTempStruct(brightness_temp=p(bt))
        ts = S.TsBoxStruct()
        t = {"z": z}
        t["perturb_ms"] = timed(lambda: lib.ComputePerturbedField(z, C.byref(icss), C.byref(pf)))
        t["halobox_ms"] = timed(lambda: lib.ComputeHaloBox(z, C.byref(icss), None, None, None, C.byref(hb)))
        t["ionize_ms"] = timed(lambda: lib.ComputeIonizedBox(z, 0.0, C.byref(pf), C.byref(pf), C.byref(prevb),
                                                             C.byref(ts), C.byref(hb), C.byref(icss), C.byref(box)))
        t["brightness_ms"] = timed(lambda: lib.ComputeBrightnessTemp(z, C.byref(ts), C.byref(box), C.byref(pf),
                                                                     C.byref(btb)))
        t["global_xH"] = float(xH.mean())
        t["mean_Tb_mK"] = float(bt.mean())
    res["snapshots"].append({k: (round(v, 3) if isinstance(v, float) else v) for k, v in t.items()})
print(json.dumps(res))
