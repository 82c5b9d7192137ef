"""Measured single-GPU pieces of a SHARDED ComputeTsBox (config 5's budget): for a world of W
ranks, the busiest rank's phase 1 (c21cm_ts_box_shard_sums: density filter loop of its shells, host tables --
frequency integrals of ITS shells only since round 6 --, box means, shell loop) and phase 2 on its cell slab
(c21cm_ts_box_shard_finish), next to the unsharded call.  The exchanges between them are bytes over links and are
not timed here.  GPU box only.

    python tools/time_ts_shard_pieces.py [HII_DIM] [WORLD] [N_THREADS]
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
import ts_helpers as H  # noqa: E402
from test_gpu_abi import Session  # noqa: E402
from test_gpu_ts_shard import FIELDS, declare, fp  # noqa: E402

S = importlib.import_module("21cmfast_amd.structs")
pkg = importlib.import_module("21cmfast_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
threads = int(sys.argv[3]) if len(sys.argv) > 3 else 16
lib = pkg.load(require_gpu=True)
declare(lib)
ses = Session(lib, pathlib.Path(tempfile.mkdtemp()), data_dir=root / "tests/golden/reference/_data", HII_DIM=n,
              DIM=2 * n, BOX_LEN=1.5 * n, SOURCE_MODEL=1, USE_TS_FLUCT=True, USE_LYA_HEATING=False, Z_HEAT_MAX=30.0,
              N_THREADS=threads)
lib.init_heat.restype = C.c_int
assert lib.init_heat() == 0
rng = np.random.default_rng(11)
shape = (n, n, n)
density = H.smooth_field(shape, rng, 0.3)
d = {"density": torch.from_numpy(density).cuda(),
     "xray_ionised_fraction": torch.from_numpy(np.exp(rng.uniform(np.log(1.5e-4), np.log(4e-4), shape)).astype(np.float32)).cuda(),
     "kinetic_temp_neutral": torch.from_numpy((9.0 * (1 + 0.6 * density)).astype(np.float32)).cuda(),
     "spin_temperature": torch.full(shape, 30.0, dtype=torch.float32, device="cuda")}
z, prev_z, ntot = 14.0, 14.3, n ** 3
pf = S.PerturbedFieldStruct(density=fp(d["density"]))
prevs = S.TsBoxStruct(**{k: fp(d[k]) for k in FIELDS})
out = {k: torch.zeros(shape, dtype=torch.float32, device="cuda") for k in FIELDS}
outs = S.TsBoxStruct(**{k: fp(v) for k, v in out.items()})


def timed(fn, reps=3):
    best = None
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        assert fn() == 0, lib.c21cm_last_error()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) * 1e3
        best = dt if best is None else min(best, dt)
    return best


t_all = timed(lambda: lib.ComputeTsBox(z, prev_z, z, 0, C.byref(pf), None, C.byref(prevs), None, C.byref(outs)))
part = torch.zeros((6, ntot), dtype=torch.float64, device="cuda")
rows = C.c_int(0)
t_sums = {}
for rank in (0, world - 1):
    t_sums[rank] = timed(lambda: lib.c21cm_ts_box_shard_sums(z, prev_z, z, C.byref(pf), C.byref(prevs), rank, world,
                                                             C.c_void_p(part.data_ptr()), C.byref(rows)))
b, e = lib.c21cm_ts_slab_begin(ntot, world, 0), lib.c21cm_ts_slab_begin(ntot, world, 1)
slab = part[:4, b:e].contiguous()
t_fin = timed(lambda: lib.c21cm_ts_box_shard_finish(z, prev_z, z, C.byref(pf), C.byref(prevs), C.c_void_p(slab.data_ptr()),
                                                    b, e - b, C.byref(outs)))
per_link_f64 = 4 * (e - b) * 8
print(json.dumps({"hii_dim": n, "world": world, "host_threads": threads, "single_gpu_ms": round(t_all, 2),
                  "phase1_ms_rank0": round(t_sums[0], 2), f"phase1_ms_rank{world - 1}": round(t_sums[world - 1], 2),
                  "phase2_slab_ms": round(t_fin, 2), "sums_bytes_per_link_f64": per_link_f64,
                  "sums_bytes_per_link_f32": per_link_f64 // 2, "outputs_bytes_per_link": 3 * (e - b) * 4,
                  "what": "phase 1 = filter loop of the rank's shells + host tables (its shells' frequency integrals) + "
                          "box means + shell loop; phase 2 = temperature update of the rank's cell slab"}))
