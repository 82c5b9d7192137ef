"""Windows evaluated inside pass X (round 3: fft_native.hip, FMODE 6 / 7, c21hip_wev_prepare) against
the streamed 3-D window tables (C21CM_WINDOWS=table, the round-1/2 path whose entries are the
double evaluation of filtering.c:80-104,357-361 rounded to float) and against the CPU oracle.

The in-kernel evaluation interpolates W(x = kR) from node tables in fp32: within 1.2e-7 of the
window's envelope of the double evaluation (tools/window_interp_check.py, restated in
test_node_table_interpolation_accuracy below on the CPU side of this file's logic).  Filtered
fields therefore agree to float round-off, and the discontinuous barrier flips only cells that
sit within ~1e-7 of it."""

import importlib
import sys
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

W = importlib.import_module("21cmfast_amd.workloads")
S = importlib.import_module("21cmfast_amd.structs")
sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "tools"))


@pytest.fixture(scope="module")
def api(gpu_lib):
    return importlib.import_module("21cmfast_amd.grid_api")


def ionize(api, spec, density, n_ion=None, **kw):
    import torch

    d = torch.from_numpy(density).cuda()
    s = None if n_ion is None else torch.from_numpy(n_ion).cuda()
    kw = {k: torch.from_numpy(v).cuda() for k, v in kw.items()}
    buf, box, rep = api.ionize_grids(spec, d, s, **kw)
    torch.cuda.synchronize()
    return buf.neutral_fraction.cpu().numpy(), buf.z_reion.cpu().numpy(), rep


@pytest.mark.parametrize("n,nz", [(128, None), (256, None), (256, 128)])
def test_lagrangian_loop_evaluated_windows_equal_tables(api, monkeypatch, n, nz):
    """G = 2 (top-hat + exp-MFP), two radii per sweep: same ionised cells, same x_HI."""
    spec = W.ionize_spec(n, hii_dim_z=nz, r_bubble_max=30.0)
    shape = (n, n, nz or n)
    density = W.density_field_numpy(shape if nz else n, seed=99)
    n_ion = W.nion_from_density(density)
    monkeypatch.setenv("C21CM_WINDOWS", "table")
    xt, zt, rt = ionize(api, spec, density, n_ion)
    monkeypatch.delenv("C21CM_WINDOWS")
    xe, ze, re = ionize(api, spec, density, n_ion)
    mismatch = np.mean((xt == 0) != (xe == 0))
    assert mismatch <= 1e-5, mismatch  # a handful of cells within ~1e-7 of the barrier
    same = (xt == 0) == (xe == 0)
    np.testing.assert_allclose(xe[same], xt[same], rtol=0, atol=3e-6)
    np.testing.assert_array_equal(ze[same], zt[same])
    assert abs(re.global_xH - rt.global_xH) < 5e-7
    n_r = spec.n_radii
    np.testing.assert_allclose(np.array(re.f_coll_grid_mean[:n_r]), np.array(rt.f_coll_grid_mean[:n_r]),
                               rtol=2e-7)
    assert 0.05 < np.mean(xe == 0) < 0.95


def test_lagrangian_loop_evaluated_windows_vs_oracle(api, oracle):
    """128^3 is the smallest box whose pass X evaluates its windows: straight against the oracle."""
    n = 128
    spec = W.ionize_spec(n, r_bubble_max=25.0)
    density = W.density_field_numpy(n, seed=4242)
    n_ion = W.nion_from_density(density)
    ref = oracle.ionize_grids(spec, density, n_ion)
    x, z, rep = ionize(api, spec, density, n_ion)
    mismatch = np.mean((x == 0) != (ref["neutral_fraction"] == 0))
    assert mismatch <= 2e-4
    same = (x == 0) == (ref["neutral_fraction"] == 0)
    np.testing.assert_allclose(x[same], ref["neutral_fraction"][same], rtol=1e-4, atol=5e-6)
    assert rep.global_xH == pytest.approx(ref["report"].global_xH, abs=2e-4)


@pytest.mark.parametrize("hii_filter", [1, 0])
def test_eulerian_loop_evaluated_windows_equal_tables(api, monkeypatch, hii_filter):
    """G = 1, closed-form f_coll (sharp-k in the reference's `simple` / `const-zeta` templates; the
    top-hat as well): same ionised cells with evaluated windows and with tables."""
    n = 128
    spec = W.ionize_spec(n, mode=W.FCOLL_ERFC, r_bubble_max=20.0)
    spec.hii_filter = hii_filter
    density = W.density_field_numpy(n, seed=7)
    monkeypatch.setenv("C21CM_WINDOWS", "table")
    xt, zt, rt = ionize(api, spec, density)
    monkeypatch.delenv("C21CM_WINDOWS")
    xe, ze, re = ionize(api, spec, density)
    assert np.mean((xt == 0) != (xe == 0)) <= 1e-5
    assert abs(re.global_xH - rt.global_xH) < 1e-6
    assert 0.02 < np.mean(xe == 0) < 0.98


@pytest.mark.parametrize("filter_type", [0, 1])
def test_ts_density_filter_loop_evaluated_windows(api, oracle, monkeypatch, filter_type):
    """fill_Rbox_table at a size whose pass X evaluates its windows, two radii per sweep (and an odd
    radius count): against the table path and against the oracle."""
    import torch

    shape, box_len = (128, 128, 128), 192.0
    rng = np.random.default_rng(3)
    f = (0.4 * rng.standard_normal(shape)).astype(np.float32)
    x = np.arange(shape[0])[:, None, None] / shape[0]
    f = (f + np.sin(2 * np.pi * 3 * x)).astype(np.float32)
    radii = [0.5, 1.2, 2.0, 3.7, 6.0, 11.0, 19.0]  # one unfiltered + six filtered... and 7 is odd
    spec = S.rbox_spec(shape[0], box_len, radii, filter_type=filter_type, min_value=-1.0,
                       const_factor=0.31)
    want = oracle.fill_Rbox_grids(spec, f)
    d = torch.from_numpy(f).cuda()
    monkeypatch.setenv("C21CM_WINDOWS", "table")
    tab = api.fill_Rbox_grids(spec, d)
    monkeypatch.delenv("C21CM_WINDOWS")
    ev = api.fill_Rbox_grids(spec, d)
    monkeypatch.setenv("C21CM_PAIR_RADII", "0")
    ev1 = api.fill_Rbox_grids(spec, d)
    scale = np.abs(want["result"]).max()
    got, gt, g1 = (r["result"].cpu().numpy() for r in (ev, tab, ev1))
    np.testing.assert_allclose(got, gt, rtol=0, atol=3e-6 * scale)
    np.testing.assert_array_equal(got, g1)  # pair sweep == single sweeps, bit for bit
    np.testing.assert_allclose(got, want["result"], rtol=2e-5, atol=2e-5 * scale)
    np.testing.assert_allclose(ev["average"], want["average"], rtol=1e-6, atol=1e-7)


def test_node_table_interpolation_accuracy():
    """The numpy emulation of weval_one (tools/window_interp_check.py): <= 1.5e-7 of the envelope."""
    import window_interp_check as C

    h, nmax = 0.25, 600
    rng = np.random.default_rng(5)
    x0 = np.concatenate([rng.uniform(0, 4, 20000), rng.uniform(0, 140, 80000)]).astype(np.float32)
    env = np.minimum(1.0, 3 / np.maximum(x0.astype(np.float64), 1e-9) ** 2)
    tab = C.node_tables(C.tophat, h, nmax)
    err = np.abs(C.interp32(tab, x0, h).astype(np.float64)
                 - C.tophat(x0.astype(np.longdouble)).astype(np.float64)) / env
    assert err.max() < 1.5e-7
    for ratio in (27.4, 2.5, 0.67):
        tab = C.node_tables(lambda x: C.expmfp(x, ratio), h, nmax)  # noqa: B023
        ex = C.expmfp(x0.astype(np.longdouble), ratio).astype(np.float64)
        assert np.abs(C.interp32(tab, x0, h).astype(np.float64) - ex).max() < 8e-8

