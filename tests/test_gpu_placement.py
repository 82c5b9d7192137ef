"""Placement walk of the work spectra (round 5, csrc/host/placement.c): where a buffer sits in HBM must change
the time of a launch and nothing else.

Two spectra written by one launch are 10-21 % slower when they come from the same physical region of the HBM
(profiles/r05_placement_study.txt); the second work spectrum of a two-grid sweep is
therefore chosen among candidates by timed launches.  A process reads C21CM_WS_PLACE once, so the two settings
run in processes of their own: same bits out, and the walk must have looked at candidates when it is on."""

import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu

ROOT = Path(__file__).resolve().parent.parent

WORKER = r"""
import hashlib, importlib, sys
sys.path.insert(0, sys.argv[1])
import torch
W = importlib.import_module("21cmfast_amd.workloads")
api = importlib.import_module("21cmfast_amd.grid_api")
n = 512
spec = W.ionize_spec(n, r_bubble_max=8.0)
density = W.density_field_torch(n, seed=4)
n_ion = W.nion_from_density(density)
for _ in range(2):  # the second call reuses the placed workspace
    buf, box, rep = api.ionize_grids(spec, density, n_ion)
    torch.cuda.synchronize()
    h = hashlib.sha256()
    for name in ("neutral_fraction", "z_reion", "kinetic_temperature"):
        h.update(getattr(buf, name).cpu().numpy().tobytes())
    print("HASH", h.hexdigest(), repr(rep.global_xH), spec.n_radii, api.ionize_last_loop_flags())
print("PLACEMENT", api.placement_report())
"""


_CACHE = {}


def _run(tmp_path, place):
    if place == "0" and "0" in _CACHE:  # (the plain-allocation run is the reference of three tests)
        return _CACHE["0"]
    out = _run_uncached(tmp_path, place)
    _CACHE[place] = out
    return out


def _run_uncached(tmp_path, place):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, C21CM_WS_PLACE=place, C21CM_WS_TRACE="1")
    env["C21CM_WS_PLACE_MS"] = "30000"  # (the time budget of a walk is not under test here)
    if place == "force":  # (this pytest process may hold device memory: the walk would see a second tenant and stay
        env["C21CM_WS_PLACE_GB"] = "96"  # out; forced here, with a budget that leaves room for xdist neighbours)
    p = subprocess.run([sys.executable, str(script), str(ROOT)], capture_output=True, text=True, env=env, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    hashes = [ln for ln in p.stdout.splitlines() if ln.startswith("HASH")]
    assert len(hashes) == 2 and hashes[0] == hashes[1], hashes
    return hashes[0], p.stderr


def test_placement_changes_nothing_but_time(tmp_path):
    off, err_off = _run(tmp_path, "0")
    on, err_on = _run(tmp_path, "force")
    assert on == off
    assert int(on.split()[-1]) & 33 == 33  # the fused loop with two radii per sweep: four work spectra
    assert "[place]" not in err_off
    # the walk timed at least two chunks for each of the two second work spectra (both radii of a sweep)
    chunks = [ln for ln in err_on.splitlines() if ln.startswith("[place]") and " chunk " in ln]
    assert len(chunks) >= 4, err_on[-1500:]


def test_walk_stays_out_when_the_device_is_shared(tmp_path, gpu_lib):
    """Round 6: with another process holding memory on the device -- here this
    pytest process, which allocates 1 GB first -- the default setting allocates plainly: no candidate is timed,
    nothing is held, and the report says why.  Results are the same bits as ever."""
    import torch

    hold = torch.zeros(1 << 28, dtype=torch.float32, device="cuda")  # 1 GB in THIS process
    torch.cuda.synchronize()
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, C21CM_WS_TRACE="1", C21CM_WS_PLACE="1")  # opted in -- but not alone on the device
    p = subprocess.run([sys.executable, str(script), str(ROOT)], capture_output=True, text=True, env=env, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    rep = eval([ln for ln in p.stdout.splitlines() if ln.startswith("PLACEMENT")][-1][len("PLACEMENT "):])
    if rep["tenants"] < 0:
        pytest.skip("the KFD's per-process accounting is not readable on this box")
    assert rep["tenants"] >= 2 and rep["outcome"] == "other tenants on the device" and rep["held_GB"] == 0
    assert " chunk " not in p.stderr
    off, _ = _run(tmp_path, "0")
    assert [ln for ln in p.stdout.splitlines() if ln.startswith("HASH")][0] == off
    del hold


def test_the_walk_is_opt_in(tmp_path):
    """The library's default allocates plainly (round 6: what a walk costs is erratic -- 0.01 to 5 s -- and is the
    caller's decision): no candidate timed, nothing held, same bits."""
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, C21CM_WS_TRACE="1")
    env.pop("C21CM_WS_PLACE", None)
    p = subprocess.run([sys.executable, str(script), str(ROOT)], capture_output=True, text=True, env=env, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    rep = eval([ln for ln in p.stdout.splitlines() if ln.startswith("PLACEMENT")][-1][len("PLACEMENT "):])
    assert rep["outcome"] == "off / not applicable" and rep["held_GB"] == 0 and rep["probes"] == 0
    assert "[place]" not in p.stderr
    off, _ = _run(tmp_path, "0")
    assert [ln for ln in p.stdout.splitlines() if ln.startswith("HASH")][0] == off


def test_two_benches_at_once_on_one_gpu():
    """Two bench.py processes started together on the one GPU both finish with the same global x_HI (the walk of
    round 5 could take three quarters of the free memory in each of them at the same moment)."""
    import json

    cmd = [sys.executable, str(ROOT / "bench.py"), "--hii-dim", "512", "--steps", "3", "--warmup", "1",
           "--no-cpu-baseline", "--no-kernel-roofline", "--no-abi"]
    env = dict(os.environ)
    env.pop("C21CM_WS_PLACE", None)
    procs = [subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env) for _ in range(2)]
    outs = [p.communicate(timeout=900) for p in procs]
    lines = []
    for p, (out, err) in zip(procs, outs):
        assert p.returncode == 0, err[-2000:]
        lines.append(json.loads(out.strip().splitlines()[-1]))
    assert lines[0]["config"]["global_xH"] == lines[1]["config"]["global_xH"]
    for ln in lines:
        assert ln["placement"]["held_GB"] <= 0.75 * 288
