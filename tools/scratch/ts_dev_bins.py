"""Per-bin deviation of the TS fixture powers and of the global history (GPU box only)."""
import importlib, sys, tempfile
from pathlib import Path
root = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(root)); sys.path.insert(0, str(root / "tests"))
import numpy as np
import test_gpu_reference_fixtures_ts as T
RP = T.RP
lib = importlib.import_module("21cmfast_amd").load(require_gpu=True)
api = importlib.import_module("21cmfast_amd.grid_api")
np.set_printoptions(precision=1, linewidth=250, suppress=True)
with tempfile.TemporaryDirectory() as d:
    got = T.evolve(lib, api, Path(d))
f = RP.fixture("power_spectra", "ts")
for k in T.TS + ("brightness_temp", "neutral_fraction"):
    p, _ = RP.get_power(got[k], RP.BOX_LEN)
    print(k, (p / f[f"coeval/power_{k}"] - 1) * 1e5)
gb = np.array([h[1] for h in got["history"]]); gx = np.array([h[2] for h in got["history"]])
print("z", np.array([h[0] for h in got["history"]]))
print("global dT_b dev e5", (gb / f["lightcone/global_brightness_temp"] - 1) * 1e5)
print("global x_HI dev e7", (gx / f["lightcone/global_neutral_fraction"] - 1) * 1e7)
for key in sorted(k for k in f.keys() if k.startswith("lightcone/global")):
    print(key)
