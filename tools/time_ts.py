"""Wall time of ComputeTsBox on the GPU box (diagnostic): E-INTEGRAL (filter loop + SFRD tables +
cell sweep) and L-INTEGRAL (cell sweep over given source grids), host arrays in and out, for a
few N_THREADS of the host-table loops.  usage: python tools/time_ts.py [HII_DIM]"""
import ctypes as C
import importlib
import os
import sys
import tempfile
import time
from pathlib import Path

root = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(root))
sys.path.insert(0, str(root / "tests"))
import numpy as np  # noqa: E402

from test_gpu_abi import Session  # noqa: E402

pkg = importlib.import_module("21cmfast_amd")
S = pkg.structs
lib = pkg.load(require_gpu=True)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
data = root / "tests" / "golden" / "reference" / "_data"
os.environ["C21CM_TS_TIMING"] = "1"
fp = lambda a: a.ctypes.data_as(S.c_float_p)  # noqa: E731
lib.ComputeTsBox.restype = C.c_int
lib.ComputeTsBox.argtypes = [C.c_float, C.c_float, C.c_float, C.c_short] + [C.c_void_p] * 5
rng = np.random.default_rng(1)
shape = (n, n, n)
density = (0.3 * rng.standard_normal(shape)).astype(np.float32)
prev = {"xray_ionised_fraction": np.full(shape, 3e-4, np.float32),
        "kinetic_temp_neutral": np.full(shape, 12.0, np.float32),
        "spin_temperature": np.full(shape, 25.0, np.float32)}
out = {k: np.zeros(shape, np.float32) for k in prev}
for model in (1, 2):
    for nt in (1, 16, 64):
        ses = Session(lib, Path(tempfile.mkdtemp()), data_dir=data, HII_DIM=n, DIM=2 * n,
                      BOX_LEN=1.5 * n, SOURCE_MODEL=model, USE_TS_FLUCT=True, USE_LYA_HEATING=False,
                      N_THREADS=nt)
        src = None
        if model == 2:
            g = (1e-3 * rng.random((40,) + shape)).astype(np.float32)
            keep = g
            src = S.XraySourceBoxStruct(filtered_sfr=fp(g), filtered_xray=fp(g))
        pf = S.PerturbedFieldStruct(density=fp(density))
        prevs = S.TsBoxStruct(**{k: fp(v) for k, v in prev.items()})
        outs = S.TsBoxStruct(**{k: fp(v) for k, v in out.items()})
        for rep in range(3):
            t0 = time.perf_counter()
            st = lib.ComputeTsBox(12.0, 12.3, 12.0, 0, C.byref(pf), C.byref(src) if src else None,
                                  C.byref(prevs), None, C.byref(outs))
            dt = time.perf_counter() - t0
            assert st == 0, lib.c21cm_last_error()
        print(f"HII_DIM={n} SOURCE_MODEL={model} N_THREADS={nt}: ComputeTsBox {1e3 * dt:.1f} ms", flush=True)
        del ses
