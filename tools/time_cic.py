#!/usr/bin/env python
"""Launch time of the PerturbedField mass deposit (c21hip_cic_scatter) for its three
implementations (C21CM_CIC = cell | tiled | direct) on synthetic fields, with the direct
global-atomic kernel as the arbiter of the result.  usage: time_cic.py [HII_DIM DIM [rms_cells]]

The displacement field is smooth (a few long modes plus small-scale noise), rms `rms_cells`
output cells per axis, like a z ~ 7 Zel'dovich field on 1.5 Mpc cells."""
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

hii, dim = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (256, 512)
rms = float(sys.argv[3]) if len(sys.argv) > 3 else 0.8
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(7)


def smooth_field(n, amp):
    x = torch.arange(n, device=dev, dtype=torch.float32) * (2 * torch.pi / n)
    f = torch.zeros((n, n, n), device=dev)
    for (a, b, c, ph) in ((1, 2, 0, 0.3), (3, 1, 2, 1.1), (2, 5, 7, 2.0), (9, 4, 1, 0.7), (17, 13, 11, 4.0)):
        f += torch.sin(a * x[:, None, None] + b * x[None, :, None] + c * x[None, None, :] + ph) / (a + b + c) ** 0.5
    f += 0.15 * torch.randn((n, n, n), device=dev, generator=g)
    return (f * (amp / f.std())).contiguous()


box_len = 1.5 * hii
growth, init_growth = 0.127, 0.0042
# displacement in output cells = v * (growth - init) / box_len * DIM * (hii / dim)
vscale = rms / ((growth - init_growth) / box_len * hii)
vel = [smooth_field(hii, vscale) for _ in range(3)]
vel2 = [smooth_field(hii, 0.1 * vscale * (growth - init_growth) / (3 / 7 * (growth**2 - init_growth**2)))
        for _ in range(3)]
dens = (2.0 * torch.randn((dim, dim, dim), device=dev, generator=g)).contiguous()
out = torch.zeros((hii, hii, hii), device=dev, dtype=torch.float64)

I3 = C.c_int * 3
P3 = C.c_void_p * 3
lib.c21hip_cic_scatter.restype = C.c_int
lib.c21hip_cic_scatter.argtypes = [C.c_void_p, I3, P3, P3, I3, C.c_void_p, I3, C.c_double, C.c_double,
                                   C.c_double, C.c_double, C.c_int, C.POINTER(C.c_int), C.c_void_p]
FIXED = os.environ.get("C21CM_CIC_ACC", "fixed")[:1] != "d"  # 64-bit fixed-point accumulation (round 5)


def run(lpt2):
    st = lib.c21hip_cic_scatter(dens.data_ptr(), I3(dim, dim, dim), P3(*[v.data_ptr() for v in vel]),
                                P3(*[v.data_ptr() for v in vel2]), I3(hii, hii, hii), out.data_ptr(),
                                I3(hii, hii, hii), box_len, box_len, growth, init_growth, lpt2,
                                C.byref(run.fixed) if FIXED else None,
                                torch.cuda.current_stream().cuda_stream)
    assert st == 0, importlib.import_module("21cmfast_amd._lib").last_error()


run.fixed = C.c_int(0)


def timed(mode, lpt2, reps=5, extra=None):
    os.environ["C21CM_CIC"] = mode
    for k, v in (extra or {}).items():
        os.environ[k] = v
    out.zero_()
    run(lpt2)
    torch.cuda.synchronize()
    # (the cell kernel may have left 2^44 fixed-point integers in the grid)
    res = out.view(torch.int64).double() / 2.0**44 if run.fixed.value else out.clone()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        run(lpt2)
    e1.record()
    torch.cuda.synchronize()
    for k in (extra or {}):
        os.environ.pop(k, None)
    return e0.elapsed_time(e1) / reps, res


report = {"hii_dim": hii, "dim": dim, "rms_cells": rms, "particles": dim**3}
for lpt2 in (1, 0):
    t_ref, ref = timed("direct", lpt2, reps=2)
    row = {"direct_ms": t_ref}
    for mode in ("tiled", "cell"):
        t, res = timed(mode, lpt2)
        err = float((res - ref).abs().max() / ref.abs().max())
        row[mode + "_ms"] = t
        row[mode + "_max_rel_err"] = err
    for brick in os.environ.get("CIC_BRICKS", "").split(";"):
        if brick:
            t, res = timed("cell", lpt2, extra={"C21CM_CIC_BRICK": brick})
            row["cell_" + brick + "_ms"] = t
            row["cell_" + brick + "_err"] = float((res - ref).abs().max() / ref.abs().max())
    for halo in os.environ.get("CIC_HALOS", "").split(";"):
        if halo:
            t, res = timed("cell", lpt2, extra={"C21CM_CIC_HALO": halo})
            row["cell_halo" + halo + "_ms"] = t
            row["cell_halo" + halo + "_err"] = float((res - ref).abs().max() / ref.abs().max())
    for diag in os.environ.get("CIC_DIAGS", "").split(";"):
        if diag:
            t, _ = timed("cell", lpt2, extra={"C21CM_CIC_DIAG": diag})
            row["cell_diag" + diag + "_ms"] = t
    row["mass_check"] = float(ref.sum() / dim**3)
    report["lpt2" if lpt2 else "zeldovich"] = row
print(json.dumps(report))
