#!/usr/bin/env python
"""InitialConditions with the reference's own random stream (rng_stream = GSL, N_THREADS streams):
wall time of one c21cm_ics_grids call with the deviates computed on the host (C21CM_GSL_DEVIATES=host)
and with the raw accepted words staged while they are drawn and turned into deviates on the device
(default), and how far the two universes differ.  usage: time_ic_gsl.py [HII_DIM DIM [N_THREADS]]"""
import importlib, json, os, sys, time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tests"))
import torch

api = importlib.import_module("21cmfast_amd.grid_api")
from test_oracle_ics import ics_spec

hii, dim = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (256, 512)
nthr = int(sys.argv[3]) if len(sys.argv) > 3 else 16
spec = ics_spec(dim, hii, box_len=1.5 * hii, seed=12345)
spec.rng_stream, spec.rng_threads = 1, nthr
res = {"hii_dim": hii, "dim": dim, "n_threads": nthr}
fields = {}
for mode in ("device", "host"):
    if mode == "host":
        os.environ["C21CM_GSL_DEVIATES"] = "host"
    ics = api.ics_grids(spec, device="cuda")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ics = api.ics_grids(spec, ics, device="cuda")
    torch.cuda.synchronize()
    res[f"ics_ms_deviates_on_{mode}"] = (time.perf_counter() - t0) * 1e3
    fields[mode] = {k: v.clone() for k, v in ics.items()}
os.environ.pop("C21CM_GSL_DEVIATES", None)
res["max_abs_diff_over_max"] = {k: float((fields["device"][k] - fields["host"][k]).abs().max() /
                                         fields["host"][k].abs().max()) for k in fields["host"]}
print(json.dumps(res))
