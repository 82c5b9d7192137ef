"""Deviation of the coeval fixtures' x_HI / dT_b power from the reference's, with the converged
sigma(M) spline and with the restated float table (C21CM_SIGMA_TABLE=reference).  GPU box only."""
import importlib
import os
import sys
import tempfile
from pathlib import Path

root = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(root))
sys.path.insert(0, str(root / "tests"))
import numpy as np  # noqa: E402

import test_gpu_reference_fixtures as T  # noqa: E402

RP = T.RP
lib = importlib.import_module("21cmfast_amd").load(require_gpu=True)
api = importlib.import_module("21cmfast_amd.grid_api")
for mode in ("spline", "reference"):
    os.environ["C21CM_SIGMA_TABLE"] = mode
    for name in ("simple", "no-mdz", "fixed_halogrids"):
        with tempfile.TemporaryDirectory() as d:
            got = T.run_coeval_abi(lib, api, Path(d), name)
        f = RP.fixture("power_spectra", name)
        out = []
        for key, arr in (("power_neutral_fraction", got["neutral_fraction"]),
                         ("power_brightness_temp", got["brightness_temp"]),
                         ("power_z_reion", got["z_reion"])):
            p, _ = RP.get_power(arr, RP.BOX_LEN)
            ref = f["coeval/" + key]
            out.append(f"{key.split('_', 1)[1]} {np.abs(p / ref - 1).max():.2e}")
        gx = got["neutral_fraction"].mean() / f["lightcone/global_neutral_fraction"][-1] - 1
        print(mode, name, *out, f"global_xH {gx:.2e}", flush=True)
