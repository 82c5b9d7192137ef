"""Wall time of E-INTEGRAL and L-INTEGRAL evolutions with and without USE_MINI_HALOS through
drivers.run_coeval, arrays resident on the device (diagnostic; GPU box only).

    PYTHONPATH=. python tools/time_coeval_mini.py [HII_DIM] [z_end] [N_THREADS]
"""
import importlib
import json
import pathlib
import sys
import time

root = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(root))
D = importlib.import_module("21cmfast_amd.drivers")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
z_end = float(sys.argv[2]) if len(sys.argv) > 2 else 8.0
n_threads = int(sys.argv[3]) if len(sys.argv) > 3 else 16
DATA = root / "tests" / "golden" / "reference" / "_data"
common = dict(HII_DIM=n, DIM=2 * n, BOX_LEN=1.5 * n, N_THREADS=n_threads, ZPRIME_STEP_FACTOR=1.04,
              Z_HEAT_MAX=30.0, HII_FILTER=0, USE_EXP_FILTER=False, CELL_RECOMB=False,
              USE_LYA_HEATING=False, SOURCE_MODEL=1, USE_TS_FLUCT=True, R_BUBBLE_MAX=30.0,
              M_TURN=10 ** 5.0)
out = {"hii_dim": n, "z_end": z_end, "n_threads": n_threads}
MINI = dict(USE_MINI_HALOS=True, ALPHA_STAR_MINI=0.5, V_CB_MODEL=3)
LAG = dict(SOURCE_MODEL=2, HII_FILTER=0, USE_EXP_FILTER=True, CELL_RECOMB=True, PERTURB_ON_HIGH_RES=False)
for label, extra in (("e_integral", {}), ("e_integral_mini", MINI), ("l_integral", LAG),
                     ("l_integral_mini", {**LAG, **MINI})):
    marks = []
    t0 = time.perf_counter()
    res = D.run_coeval(D.Inputs(random_seed=3, **{**common, **extra}), [z_end], data_path=DATA,
                       device="cuda", keep=("neutral_fraction",),
                       progress=lambda h: marks.append(time.perf_counter()))
    total = time.perf_counter() - t0
    steps = [1e3 * (b - a) for a, b in zip(marks[:-1], marks[1:])]
    out[label] = {"snapshots": len(marks), "total_s": round(total, 2),
                  "ms_per_snapshot_median": round(sorted(steps)[len(steps) // 2], 1),
                  "ms_per_snapshot_last": round(steps[-1], 1),
                  "x_HI_end": round(res["history"][-1][2], 4)}
print(json.dumps(out))
