import sys, importlib, ctypes as C
from pathlib import Path
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import torch, numpy as np
pkg = importlib.import_module("21cmfast_amd"); lib = pkg.load(require_gpu=True)
S = importlib.import_module("21cmfast_amd.structs")
from test_gpu_ts_shard import setup, declare, fp, FIELDS
declare(lib)
import tempfile
for sm in (1, 0):
    ses, d = setup(lib, Path(tempfile.mkdtemp()), 64, sm)
    pf = S.PerturbedFieldStruct(density=fp(d["density"]))
    prevs = S.TsBoxStruct(**{k: fp(d[k]) for k in FIELDS})
    rows = C.c_int(0)
    for world in (1, 3):
        for rank in range(world):
            part = torch.zeros((6, 64**3), dtype=torch.float64, device="cuda")
            assert lib.c21cm_ts_box_shard_sums(14.0, 14.3, 14.0, C.byref(pf), C.byref(prevs), rank, world, C.c_void_p(part.data_ptr()), C.byref(rows)) == 0
            for k in range(rows.value):
                a = part[k].abs()
                print(sm, world, rank, k, float(a.min()), float(a.max()), float(part[k].min()))
