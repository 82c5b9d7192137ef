"""Reproducibility of the 1024-point line passes while several processes share the GPU (round 5).

The race this guards against: the 1024-point line passes read their twiddle tables from LDS for the register
stage of a workgroup's FIRST tile without a barrier after the tables' load (fft_native.hip: line_pass_kernel).
In isolation the waves of a workgroup start together and nothing showed in four rounds of tests; with four
processes on the GPU about one 1024^3 call in thirty came out with a wrong x-plane.  Here four processes
transform thin boxes with 1024-point x- and z-lines over and over and every result must equal the first
(validated against a build with the bug back in: C21X_NO_TW_BARRIER=1)."""

import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu

ROOT = Path(__file__).resolve().parent.parent

WORKER = r"""
import importlib, sys
sys.path.insert(0, sys.argv[1])
import torch
W = importlib.import_module("21cmfast_amd.workloads")
api = importlib.import_module("21cmfast_amd.grid_api")
reps = int(sys.argv[2])
bad = 0
# (a) the excursion-set loop on a box with 1024-point lines along every axis of the transforms it uses
n = 1024
spec = W.ionize_spec(64, hii_dim_z=64, r_bubble_max=6.0)
# (b) plain transforms of thin boxes: 1024-point x-lines (forward and backward line passes), 1024-point z-lines
for shape in ((1024, 64, 64), (1024, 64, 1024)):
    nx, ny, nz = shape
    g = torch.Generator(device="cuda").manual_seed(7)
    pad = torch.zeros((nx, ny, nz + 2), device="cuda")
    pad[:, :, :nz] = torch.randn(shape, device="cuda", generator=g)
    first = None
    for it in range(reps):
        d = pad.clone()
        api.fft_r2c(d, nx, ny, nz)
        spec_k = d.clone()
        api.fft_c2r(d, nx, ny, nz)
        torch.cuda.synchronize()
        if first is None:
            first = (spec_k, d.clone())
        else:
            if not torch.equal(first[0], spec_k) or not torch.equal(first[1], d):
                bad += 1
print("BAD", bad)
"""


def test_line_passes_reproducible_under_contention(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ)
    procs = [subprocess.Popen([sys.executable, str(script), str(ROOT), "40"], stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True, env=env) for _ in range(4)]
    outs = [p.communicate(timeout=900) for p in procs]
    for p, (out, err) in zip(procs, outs):
        assert p.returncode == 0, err[-2000:]
        lines = [ln for ln in out.splitlines() if ln.startswith("BAD")]
        assert lines and lines[-1] == "BAD 0", (out[-500:], err[-500:])
