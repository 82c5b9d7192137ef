"""GPU parity of the spin-temperature filter stage against the CPU oracle.

fill_Rbox_table, one_annular_filter and UpdateXraySourceBox (SpinTemperatureBox.c:560-808)
through the C ABI, on the rocFFT path (sizes the native transform does not cover) and on the
native split-layout path (power-of-two sizes >= 64), host and device arrays, with the
multiple-scattering window.  Tolerance: the outputs are float32 results of a float32 FFT
pair, compared at rtol 2e-5 of the field's scale (the filter stage's own rounding differs
between the two FFT factorizations); statistics at 1e-6 relative.
"""

import ctypes as C
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

S = importlib.import_module("21cmfast_amd.structs")


@pytest.fixture(scope="module")
def api(gpu_lib):
    return importlib.import_module("21cmfast_amd.grid_api")


_KEEP = []  # Broadcast_struct_global_all stores pointers: keep the structs alive


def _broadcast(lib, n, box_len, n_step_ts=40, ms=False, mini=False):
    so = S.default_simulation_options(HII_DIM=n, DIM=2 * n, BOX_LEN=box_len)
    mo, cp, ct = S.default_matter_options(), S.default_cosmo_params(), S.default_cosmo_tables()
    ap = S.default_astro_params(N_STEP_TS=n_step_ts)
    ao = S.default_astro_options(LYA_MULTIPLE_SCATTERING=ms, USE_MINI_HALOS=mini)
    _KEEP.append((so, mo, cp, ap, ao, ct))
    lib.Broadcast_struct_global_all(C.byref(so), C.byref(mo), C.byref(cp), C.byref(ap),
                                    C.byref(ao), C.byref(ct))


def _field(shape, seed=5, positive=False):
    rng = np.random.default_rng(seed)
    f = rng.standard_normal(shape).astype(np.float32)
    # a few large-scale modes so that smoothing leaves structure
    x = np.arange(shape[0])[:, None, None] / shape[0]
    y = np.arange(shape[1])[None, :, None] / shape[1]
    f = (0.5 * f + np.sin(2 * np.pi * (x + 2 * y))).astype(np.float32)
    return (np.abs(f) + np.float32(0.05)).astype(np.float32) if positive else f


def _close(got, want, rtol=2e-5):
    scale = np.abs(want).max()
    np.testing.assert_allclose(got, want, rtol=rtol, atol=rtol * scale)


@pytest.mark.parametrize("shape,box_len", [((24, 24, 24), 36.0), ((64, 64, 64), 96.0),
                                           ((64, 64, 128), 96.0), ((64, 64, 512), 96.0),
                                           ((64, 64, 1024), 96.0)])
@pytest.mark.parametrize("filter_type", [0, 1, 2])
def test_fill_Rbox_matches_oracle(api, oracle, shape, box_len, filter_type):
    import torch

    f = (0.4 * _field(shape)).astype(np.float32)
    radii = [0.5, 1.2, 2.0, 3.7, 6.0, 11.0]
    spec = S.rbox_spec(shape[0], box_len, radii, filter_type=filter_type, min_value=-1.0,
                       const_factor=0.31, hii_dim_z=shape[2],
                       box_len_z=box_len * shape[2] / shape[0])
    want = oracle.fill_Rbox_grids(spec, f)
    got_host = api.fill_Rbox_grids(spec, f)  # host arrays: staged
    got_dev = api.fill_Rbox_grids(spec, torch.from_numpy(f).cuda())  # device arrays: in place
    for got in (got_host["result"], got_dev["result"].cpu().numpy()):
        _close(got, want["result"])
    for got in (got_host, got_dev):
        np.testing.assert_allclose(got["average"], want["average"], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(got["min"], want["min"], rtol=2e-5, atol=2e-5)
        np.testing.assert_allclose(got["max"], want["max"], rtol=2e-5, atol=2e-5)
    np.testing.assert_array_equal(got_host["result"], got_dev["result"].cpu().numpy())
    # the floor is hit (unfiltered radius) and the statistics are those of the stored values
    res = got_dev["result"].cpu().numpy().reshape(len(radii), -1)
    assert got_dev["min"][0] == pytest.approx(np.float32(-1.0) * 0.31, rel=1e-6)
    np.testing.assert_array_equal(got_dev["min"], res.min(1).astype(np.float64))
    np.testing.assert_array_equal(got_dev["max"], res.max(1).astype(np.float64))
    np.testing.assert_allclose(got_dev["average"], res.astype(np.float64).mean(1), rtol=1e-9)


@pytest.mark.parametrize("shape,box_len", [((20, 20, 20), 40.0), ((64, 64, 64), 128.0),
                                           ((128, 128, 64), 256.0)])
@pytest.mark.parametrize("types,R_star", [([4], 0.0), ([4, 4], 0.0), ([5, 4], 7.0),
                                          ([5, 4, 5, 4, 4], 3.0)])
def test_annular_filter_matches_oracle(api, oracle, shape, box_len, types, R_star):
    import torch

    grids = [_field(shape, seed=10 + i, positive=(i % 2 == 0)) for i in range(len(types))]
    grids[-1] = _field(shape, seed=99)  # zero-mean: smoothed cells below zero -> the zero floor
    grids[-1][3, 4, 5] += 40.0
    cell = box_len / shape[0]
    spec = S.annular_spec(shape[0], box_len, 2.5 * cell, 5.0 * cell, types, R_star=R_star,
                          hii_dim_z=shape[2], box_len_z=box_len * shape[2] / shape[0])
    want = oracle.annular_filter_grids(spec, grids)
    got_host = api.annular_filter_grids(spec, grids)
    got_dev = api.annular_filter_grids(spec, [torch.from_numpy(g).cuda() for g in grids])
    for g in range(len(types)):
        for got in (got_host["outputs"][g], got_dev["outputs"][g].cpu().numpy()):
            _close(got, want["outputs"][g])
            assert got.min() >= 0
        np.testing.assert_array_equal(got_host["outputs"][g], got_dev["outputs"][g].cpu().numpy())
    for got in (got_host, got_dev):
        np.testing.assert_allclose(got["u_avg"], want["u_avg"], rtol=1e-6)
        np.testing.assert_allclose(got["f_avg"], want["f_avg"], rtol=1e-5)
    assert (want["outputs"][-1] == 0).any()


def test_annular_cell_scale_shell_is_not_filtered(api):
    import torch

    a = torch.from_numpy(_field((64, 64, 64), positive=True)).cuda()
    out = api.annular_filter_grids(S.annular_spec(64, 96.0, 0.0, 1.5, [4]), [a])
    np.testing.assert_allclose(out["outputs"][0].cpu().numpy(), a.cpu().numpy(), rtol=2e-6, atol=2e-6)
    assert out["f_avg"][0] == pytest.approx(out["u_avg"][0], rel=1e-6)


def test_ms_window_limits_on_device(api):
    """tests/test_filtering.py:326-366 on the device: the multiple-scattering window equals
    the straight-line shell for R_star -> 0 (atol 1e-4) and differs otherwise."""
    import torch

    rng = np.random.default_rng(12345)
    box = torch.from_numpy(rng.random((64, 64, 64)).astype(np.float32)).cuda()
    for R_star in (1e-6, 5.0, 20.0):
        sl = api.annular_filter_grids(S.annular_spec(64, 128.0, 4.0, 12.0, [4], R_star), [box])
        ms = api.annular_filter_grids(S.annular_spec(64, 128.0, 4.0, 12.0, [5], R_star), [box])
        a, b = sl["outputs"][0].cpu().numpy(), ms["outputs"][0].cpu().numpy()
        if R_star < 1:
            np.testing.assert_allclose(a, b, atol=1e-4)
        else:
            assert not np.allclose(a, b, atol=1e-4)
        assert ms["f_avg"][0] == pytest.approx(ms["u_avg"][0], rel=1e-5)


def test_test_filter_hook_with_the_ms_window(api, oracle, pkg):
    """The exported test hook (filtering.c:397-445) with filter_flag 5."""
    lib = pkg.load()
    n, box_len = 32, 64.0
    _broadcast(lib, n, box_len)
    rng = np.random.default_rng(3)
    box = rng.random((n, n, n)).astype(np.float32)
    res = np.zeros((n, n, n), np.float64)
    st = lib.test_filter(box.ctypes.data_as(C.c_void_p), 4.0, 10.0, 6.0, 5,
                         res.ctypes.data_as(C.c_void_p))
    assert st == 0, lib.c21cm_last_error()
    # oracle: the annular filter without its zero floor == the filter primitive (positive box)
    want = oracle.annular_filter_grids(S.annular_spec(n, box_len, 4.0, 10.0, [5], 6.0), [box])
    _close(res.astype(np.float32), want["outputs"][0])


@pytest.mark.parametrize("ms,mini", [(False, False), (True, False), (True, True)])
def test_UpdateXraySourceBox_abi(api, oracle, pkg, ms, mini):
    """The ABI entry point (SpinTemperatureBox.c:748-808) with host arrays, globals broadcast
    as py21cmfast does."""
    lib = pkg.load()
    n, box_len, n_step = 64, 96.0, 4
    _broadcast(lib, n, box_len, n_step_ts=n_step, ms=ms, mini=mini)
    shape = (n, n, n)
    sfr = _field(shape, seed=1, positive=True)
    xray = _field(shape, seed=2, positive=True)
    sfr_mini = _field(shape, seed=3, positive=True)
    ntot = n**3
    out = {k: np.zeros(n_step * ntot, np.float32)
           for k in ("filtered_sfr", "filtered_xray", "filtered_sfr_mini", "filtered_sfr_lw",
                     "filtered_sfr_mini_lw")}
    means = {k: np.zeros(n_step, np.float64) for k in ("mean_log10_Mcrit_LW", "mean_sfr", "mean_sfr_mini")}
    fp = lambda a: a.ctypes.data_as(S.c_float_p)
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    hb = S.HaloBoxStruct(halo_sfr=fp(sfr), halo_xray=fp(xray), halo_sfr_mini=fp(sfr_mini),
                         log10_Mcrit_MCG_ave=5.75)
    xb = S.XraySourceBoxStruct(**{k: fp(v) for k, v in out.items()}, **{k: dp(v) for k, v in means.items()})
    R_ct, R_in, R_out, R_star = 2, 3.0, 6.5, 4.0
    st = lib.UpdateXraySourceBox(C.byref(hb), R_in, R_out, R_ct, R_star, C.byref(xb))
    assert st == 0, lib.c21cm_last_error()
    lya = 5 if ms else 4
    want = oracle.annular_filter_grids(
        S.annular_spec(n, box_len, R_in, R_out, [lya, 4, lya, 4, 4], R_star),
        [sfr, xray, sfr_mini, sfr, sfr_mini])
    sl = slice(R_ct * ntot, (R_ct + 1) * ntot)
    _close(out["filtered_sfr"][sl].reshape(shape), want["outputs"][0])
    _close(out["filtered_xray"][sl].reshape(shape), want["outputs"][1])
    assert means["mean_sfr"][R_ct] == pytest.approx(want["f_avg"][0], rel=1e-5)
    if mini:
        _close(out["filtered_sfr_mini"][sl].reshape(shape), want["outputs"][2])
        _close(out["filtered_sfr_lw"][sl].reshape(shape), want["outputs"][3])
        _close(out["filtered_sfr_mini_lw"][sl].reshape(shape), want["outputs"][4])
        assert means["mean_sfr_mini"][R_ct] == pytest.approx(want["f_avg"][2], rel=1e-5)
        assert means["mean_log10_Mcrit_LW"][R_ct] == 5.75
    else:
        assert not out["filtered_sfr_mini"].any() and means["mean_sfr_mini"][R_ct] == 0
    # only slot R_ct was written
    for k in ("filtered_sfr", "filtered_xray"):
        assert not out[k][: R_ct * ntot].any() and not out[k][(R_ct + 1) * ntot:].any()
    # bad radius index
    assert lib.UpdateXraySourceBox(C.byref(hb), R_in, R_out, n_step, R_star, C.byref(xb)) == 3


def test_filter_stage_properties_at_full_size(api):
    """Size-independent properties at 512^3 (BASELINE config 3's grid): a normalised window
    conserves the box mean of a positive field; radii at or below the cell scale return the
    input; the variance falls monotonically with the radius."""
    import torch

    n, box_len = 512, 768.0
    g = torch.Generator(device="cuda").manual_seed(7)
    f = torch.rand((n, n, n), device="cuda", generator=g) + 0.25
    radii = [0.5, 3.0, 9.0, 27.0]
    out = api.fill_Rbox_grids(S.rbox_spec(n, box_len, radii, filter_type=0, min_value=-1.0,
                                          const_factor=2.0), f)
    mean = float(f.double().mean())
    np.testing.assert_allclose(out["average"], 2.0 * mean, rtol=2e-6)
    torch.testing.assert_close(out["result"][0], 2.0 * f, rtol=3e-6, atol=3e-6)
    sd = [float(out["result"][r].double().std()) for r in range(len(radii))]
    assert sd[0] > sd[1] > sd[2] > sd[3] > 0
    ann = api.annular_filter_grids(S.annular_spec(n, box_len, 6.0, 12.0, [4, 5], R_star=9.0), [f, f])
    np.testing.assert_allclose(ann["f_avg"], ann["u_avg"], rtol=2e-6)
    np.testing.assert_allclose(ann["u_avg"], mean, rtol=1e-9)
    assert float(ann["outputs"][0].min()) > 0
    assert not torch.allclose(ann["outputs"][0], ann["outputs"][1], atol=1e-4)
