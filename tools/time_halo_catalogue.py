"""Time of the halo-catalogue branch of ComputeHaloBox's grid call (c21cm_halobox_grids with a
HaloCatalog, everything resident on the device): halos only, and halos + the integrated part
(diagnostic; GPU box only).

    PYTHONPATH=. python tools/time_halo_catalogue.py [HII_DIM] [n_halos] [reps] [sorted]

sorted = 1: the catalogue in the order of its Lagrangian cells, as the reference's sampler writes
it (one cell after the other); 0: halos in random order.
"""
import ctypes as C
import importlib
import json
import pathlib
import sys
import time

import numpy as np
import torch

root = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(root))
sys.path.insert(0, str(root / "tests"))
S = importlib.import_module("21cmfast_amd.structs")
api = importlib.import_module("21cmfast_amd.grid_api")
from halo_catalogue_helpers import halo_consts, random_catalogue  # noqa: E402
from test_oracle_halobox import halobox_spec, make_tables, random_ics, with_xray  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n_halos = int(float(sys.argv[2])) if len(sys.argv) > 2 else 10_000_000
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
tables = make_tables()
in_cell_order = bool(int(sys.argv[4])) if len(sys.argv) > 4 else False
cat = random_catalogue(n_halos, 1.5 * n, seed=1)
if in_cell_order:
    cell = np.floor(cat["coords"] / 1.5).astype(np.int64) % n
    order = np.argsort((cell[:, 0] * n + cell[:, 1]) * n + cell[:, 2], kind="stable")
    cat = {k: np.ascontiguousarray(v[order]) for k, v in cat.items()}
dev = {k: torch.from_numpy(v).cuda() for k, v in cat.items()}
ics = {k: torch.from_numpy(v).cuda() for k, v in random_ics(n, n, False, seed=2, vscale=3.0).items()}
hc = S.HaloCatalogStruct(n_halos=n_halos, buffer_size=n_halos,
                         **{f: C.cast(dev[k].data_ptr(), S.c_float_p) for f, k in (
                             ("halo_masses", "masses"), ("halo_coords", "coords"), ("star_rng", "star_rng"),
                             ("sfr_rng", "sfr_rng"), ("xray_rng", "xray_rng"))})
out = {"hii_dim": n, "n_halos": n_halos, "reps": reps, "cell_order": in_cell_order}
for label, skip, halos in (("integral_only", False, False), ("halos_only", True, True), ("halos_and_integral", False, True)):
    spec = with_xray(halobox_spec(n, n, False, tables), tables)
    consts = halo_consts()
    if halos:
        spec.halos, spec.halo_consts, spec.skip_integral = C.pointer(hc), C.pointer(consts), int(skip)
    api.halobox_grids(spec, ics, with_whalo=True, with_xray=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        res = api.halobox_grids(spec, ics, with_whalo=True, with_xray=True)
    torch.cuda.synchronize()
    out[label + "_ms"] = round(1e3 * (time.perf_counter() - t0) / reps, 3)
    out[label + "_n_ion_sum"] = float(res["n_ion"].double().sum())
# the same call with the catalogue in (pageable) host memory: 28 B per halo over PCIe first
host_cat = S.halo_catalog(cat["masses"], cat["coords"], cat["star_rng"], cat["sfr_rng"], cat["xray_rng"])
spec = with_xray(halobox_spec(n, n, False, tables), tables)
consts = halo_consts()
spec.halos, spec.halo_consts = C.pointer(host_cat), C.pointer(consts)
api.halobox_grids(spec, ics, with_whalo=True, with_xray=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    api.halobox_grids(spec, ics, with_whalo=True, with_xray=True)
torch.cuda.synchronize()
out["halos_and_integral_host_catalogue_ms"] = round(1e3 * (time.perf_counter() - t0) / reps, 3)
out["halo_part_ms"] = round(out["halos_and_integral_ms"] - out["integral_only_ms"], 3)
out["halos_per_s"] = round(n_halos / (1e-3 * out["halo_part_ms"]), 0)
# three grids x 8 cells of fp64 atomics per halo (n_ion, SFR, L_X)
out["fp64_atomics_per_s"] = round(24 * n_halos / (1e-3 * out["halo_part_ms"]), 0)
print(json.dumps(out))
