"""CPU: oracle grid algorithm + the library's host scalars against the reference's coeval fixtures,
with the converged host quadratures and with C21CM_HOST_MODE=reference."""
import importlib
import os
import sys
import tempfile
from pathlib import Path

root = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(root))
sys.path.insert(0, str(root / "tests"))
import numpy as np  # noqa: E402

import refpin as RP  # noqa: E402
import test_reference_fixtures_ionize as T  # noqa: E402

pkg = importlib.import_module("21cmfast_amd")
oracle = importlib.import_module("oracle.oracle")
oracle.load()
S = T.S
spec = RP.ics_spec(2, 0, 2)
ics = oracle.ics_grids(spec, oracle.new_ics_arrays(spec))
pf = oracle.perturb_grids(RP.perturb_spec(T.Z), ics)
for mode in ("cc", "rc", "cr", "rr"):
    os.environ["C21CM_SIGMA_TABLE"] = mode[0]
    os.environ["C21CM_MF_QUAD"] = mode[1]
    for name, sm in (("simple", 1), ("no-mdz", 0)):
        with tempfile.TemporaryDirectory() as d:
            ses = T.session(pkg, Path(d), sm)
            sp = T.eulerian_spec(ses, pkg.load(), oracle, sm)
            out = oracle.ionize_grids(sp, pf["density"], need_nion=True)
            f = RP.fixture("power_spectra", name)
            p_x, _ = RP.get_power(out["neutral_fraction"], RP.BOX_LEN)
            bt = oracle.brightness_grids(S.brightness_spec(pf["density"].size, T.Z, cosmo=ses.cp),
                                         pf["density"], out["neutral_fraction"])
            p_b, _ = RP.get_power(bt["brightness_temp"], RP.BOX_LEN)
            dx = p_x / f["coeval/power_neutral_fraction"] - 1
            db = p_b / f["coeval/power_brightness_temp"] - 1
            print(mode, name, f"mean_f_coll {sp.mean_f_coll:.9e}", f"x_HI power dev max {np.abs(dx).max():.2e} mean {dx.mean():+.2e}",
                  f"dT_b power dev max {np.abs(db).max():.2e} mean {db.mean():+.2e}",
                  f"global dTb {bt['mean'] / f['lightcone/global_brightness_temp'][-1] - 1:+.2e}", flush=True)
            del ses
print(RP.check_coeval_fields("simple", {"lowres_density": ics["lowres_density"], "density": pf["density"],
                                        "velocity_z": pf["velocity_z"]}))
