#!/usr/bin/env python
"""Launch time of the plane-fused pass Y + Z (plane_yz.hip) against the separate pass Y and fused
pass Z on random 512^3 work spectra, and -- with a C21X_YZ_PROF build (C21CM_LIB=variants/...) --
where a workgroup's time goes.  GPU box only."""
import ctypes as C
import importlib
import json
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

pkg = importlib.import_module("21cmfast_amd")
lib = pkg.load(require_gpu=True)
n = 512
nfl = lib.c21hip_split_floats
nfl.restype = C.c_size_t
nfl.argtypes = [C.c_int] * 3
fl = nfl(n, n, n)
g = torch.Generator(device="cuda").manual_seed(3)
wa = torch.randn(fl + 4096, device="cuda", generator=g)
wb = torch.randn(fl + 4096, device="cuda", generator=g).abs()
mask = torch.zeros(n**3, dtype=torch.uint8, device="cuda")
partials = torch.zeros(n * n // 4 + 64, dtype=torch.float64, device="cuda")
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
lib.c21hip_plane_yz_ionise.restype = C.c_int
lib.c21hip_plane_yz_ionise.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                       C.c_int, C.c_double, C.c_double, C.c_int, C.c_double, C.c_int, C.c_void_p]
lib.c21hip_split_y_nyq.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
lib.c21hip_bench_pass.restype = C.c_int
lib.c21hip_bench_pass.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_double, C.c_int,
                                  C.c_void_p, C.POINTER(C.c_float)]


def yz():
    sh = int(os.environ.get("C21CM_YZ_SHIFT", "0"))  # timing experiment: shifted base addresses (bytes)
    st = lib.c21hip_plane_yz_ionise(wa.data_ptr() + sh, wb.data_ptr() + sh, mask.data_ptr(), partials.data_ptr(), n, n, n,
                                    5, 6.2e9, 1.0, 1, 1e-9, -1, stream)
    assert st == 0


def timed(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


out = {"plane_yz_ms": timed(yz),
       "nyq_y_ms": timed(lambda: lib.c21hip_split_y_nyq(wa.data_ptr(), wb.data_ptr(), n, n, n, stream))}
for kind, name in ((1, "pass_y_ms"), (2, "pass_z_fused_ms")):
    ms = C.c_float()
    lib.c21hip_bench_pass(kind, n, 0, 3, 10.0, 25.0, 768.0, 20, stream, C.byref(ms))
    out[name] = ms.value
lib.c21hip_plane_yz_status.argtypes = [C.c_void_p]
out["status_flags"] = lib.c21hip_plane_yz_status(stream)
if hasattr(lib, "c21hip_plane_yz_profile"):
    buf = (C.c_ulonglong * 2048)()
    if lib.c21hip_plane_yz_profile(buf) == 0:
        t = torch.tensor(list(buf), dtype=torch.float64).view(256, 8) * 1e-2 / 64  # us per plane
        names = ["y_compute", "wait_slot_free", "store_arrive", "wait_plane", "rowload_arrive", "z_compute"]
        if os.environ.get("C21CM_YZ") == "2":  # role-split kernel: Y role 0-3, Z role (a density wave) 4-7
            names = ["y_compute", "y_wait_slot_free", "y_store_arrive", "-", "z_wait_plane", "z_rowload_arrive",
                     "z_c2r", "z_exchange_barrier_mask_sum"]
        out["us_per_plane_mean"] = {nm: round(float(t[:, i].mean()), 3) for i, nm in enumerate(names)}
        out["us_per_plane_max"] = {nm: round(float(t[:, i].max()), 3) for i, nm in enumerate(names)}
        if os.environ.get("C21CM_YZ_DUMP"):  # [k][xcd] of one phase (blockIdx = 8 k + xcd)
            col = int(os.environ["C21CM_YZ_DUMP"])
            out["dump"] = [[round(float(v), 2) for v in row] for row in t[:, col].view(32, 8)]
        out["us_per_plane_total_mean"] = round(float(t.sum(dim=1).mean()), 3)
        out["us_per_plane_roles"] = [round(float(t[:, :3].sum(dim=1).mean()), 3), round(float(t[:, 4:].sum(dim=1).mean()), 3)]
print(json.dumps(out))
