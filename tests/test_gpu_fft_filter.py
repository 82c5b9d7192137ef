"""GPU parity: device FFT and k-space filters vs the CPU oracle (through the C ABI)."""

import importlib

import numpy as np
import pytest

from test_oracle_fft_filter import HII_DIM, BOX_LEN, delta_function_checks

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def api(gpu_lib):
    return importlib.import_module("21cmfast_amd.grid_api")


@pytest.mark.parametrize("shape", [(16, 16, 16), (35, 35, 35), (50, 50, 50), (64, 64, 64),
                                   (70, 70, 35), (128, 128, 128), (150, 150, 150),
                                   # 3 * 2^L lines of the default DIM = 3 HII_DIM grids: native
                                   # passes with a final radix-3 stage
                                   (192, 192, 192), (384, 192, 64), (64, 384, 192),
                                   (192, 64, 768), (96, 96, 96)])
def test_fft_roundtrip_and_spectrum(api, oracle, shape):
    import torch

    nx, ny, nz = shape
    rng = np.random.default_rng(5)
    a = rng.standard_normal(shape).astype(np.float32)
    pad = np.zeros((nx, ny, 2 * (nz // 2 + 1)), np.float32)
    pad[:, :, :nz] = a
    d = torch.from_numpy(pad).cuda()
    api.fft_r2c(d, nx, ny, nz)
    spec = d.cpu().numpy().view(np.complex64).reshape(nx, ny, nz // 2 + 1)
    ref = oracle.fft_r2c(a)
    scale = np.abs(ref).max()
    assert np.abs(spec - ref).max() <= 2e-5 * scale  # float32 FFT vs float32 FFT
    api.fft_c2r(d, nx, ny, nz)
    back = d.cpu().numpy()[:, :, :nz] / (nx * ny * nz)
    np.testing.assert_allclose(back, a, atol=2e-5)


def test_radix3_sizes_run_on_the_native_passes(gpu_lib):
    """192 / 384 / 768-point lines leave rocFFT (the reference's default is DIM = 3 HII_DIM,
    wrapper/inputs.py:915), and since round 3 so do 1536-point lines (8-column tiles, one radix-2
    stage + two 768-point transforms); 96 does not (loader geometry)."""
    for n in (192, 384, 768):
        assert gpu_lib.c21hip_fft_is_native(n, n, n)
        assert gpu_lib.c21hip_fft_is_native(256, n, 64)
    assert gpu_lib.c21hip_fft_is_native(1536, 64, 1536) and gpu_lib.c21hip_fft_is_native(256, 1536, 64)
    for n in (96, 150, 320):
        assert not gpu_lib.c21hip_fft_is_native(n, n, n)


@pytest.mark.parametrize("shape", [(768, 64, 768), (1536, 64, 1536), (64, 1536, 192),
                                   # 512- and 1024-point x / y lines: the passes whose first and last
                                   # Stockham stages run on the registers (round 3), both directions,
                                   # with short and long z-lines beside them
                                   (512, 64, 128), (64, 512, 1024), (1024, 64, 64), (128, 1024, 256),
                                   (512, 1024, 64)])
def test_fft_768_and_1536_against_numpy(api, shape):
    """Full 512-, 768-, 1024- and 1536-point transforms along every axis (thin boxes) against numpy in
    double; 1536 = the reference's default DIM at HII_DIM = 512."""
    import torch

    rng = np.random.default_rng(9)
    a = rng.standard_normal(shape).astype(np.float32)
    pad = np.zeros((shape[0], shape[1], shape[2] + 2), np.float32)
    pad[:, :, :shape[2]] = a
    d = torch.from_numpy(pad).cuda()
    api.fft_r2c(d, *shape)
    spec = d.cpu().numpy().view(np.complex64).reshape(shape[0], shape[1], shape[2] // 2 + 1)
    ref = np.fft.rfftn(a.astype(np.float64))
    assert np.abs(spec - ref).max() <= 3e-5 * np.abs(ref).max()
    api.fft_c2r(d, *shape)
    back = d.cpu().numpy()[:, :, :shape[2]] / float(np.prod(shape))
    np.testing.assert_allclose(back, a, atol=3e-5)


def test_fft_linearity_large(api):
    """Size-independent property at a bench-scale box: FFT(a + 2b) = FFT(a) + 2 FFT(b)."""
    import torch

    n = 256
    g = torch.Generator(device="cuda").manual_seed(7)
    a = torch.randn((n, n, n + 2), generator=g, device="cuda")
    b = torch.randn((n, n, n + 2), generator=g, device="cuda")
    c = a + 2 * b
    for t in (a, b, c):
        api.fft_r2c(t, n, n, n)
    err = (c - (a + 2 * b)).abs().max().item()
    assert err <= 1e-4 * c.abs().max().item()


@pytest.mark.parametrize("R", [1.5, 5.0, 10.0, 20.0])
@pytest.mark.parametrize("filter_type", [0, 1, 2, 3, 4])
def test_delta_function_known_answer_device(api, filter_type, R):
    """The reference's analytic filter test (tests/test_filtering.py:111-236) on the HIP path."""
    delta_function_checks(api.filter_grid, filter_type, R)


@pytest.mark.parametrize("shape", [(50, 50, 50), (64, 64, 64), (48, 48, 96), (64, 64, 128),
                                   (128, 128, 64), (192, 192, 64), (64, 64, 192)])
@pytest.mark.parametrize("filter_type,R,R_param", [(0, 3.0, 0.0), (0, 12.0, 0.0), (1, 6.0, 0.0),
                                                   (2, 4.0, 0.0), (3, 7.5, 37.66), (4, 5.0, 9.0)])
def test_filter_matches_oracle_random_box(api, oracle, shape, filter_type, R, R_param):
    rng = np.random.default_rng(11)
    box = rng.standard_normal(shape).astype(np.float32)
    box_len = 1.5 * shape[0]
    box_len_z = box_len * shape[2] / shape[0]
    got = api.filter_grid(box, box_len, filter_type, R, R_param, box_len_z=box_len_z)
    ref = oracle.filter_grid(box, box_len, filter_type, R, R_param, box_len_z=box_len_z)
    # tolerance: float32 FFT round-off on O(1)-amplitude data (rtol 1e-5 of the field scale)
    np.testing.assert_allclose(got, ref, atol=2e-5 * np.abs(ref).max() + 1e-7, rtol=1e-4)


def test_exported_test_filter_hook(gpu_lib, oracle, pkg):
    """`test_filter` through the reference ABI: broadcast globals, float in, double out."""
    import ctypes as C

    S = pkg.structs
    n = 32
    so = S.default_simulation_options(HII_DIM=n, DIM=2 * n, BOX_LEN=48.0)
    mo, cp = S.default_matter_options(), S.default_cosmo_params()
    ap, ao, ct = S.default_astro_params(), S.default_astro_options(), S.default_cosmo_tables()
    gpu_lib.Broadcast_struct_global_all(C.byref(so), C.byref(mo), C.byref(cp), C.byref(ap),
                                        C.byref(ao), C.byref(ct))
    rng = np.random.default_rng(2)
    box = rng.standard_normal((n, n, n)).astype(np.float32)
    res = np.zeros((n, n, n), np.float64)
    st = gpu_lib.test_filter(box.ctypes.data_as(C.c_void_p), 5.0, 0.0, 0.0, 0,
                             res.ctypes.data_as(C.c_void_p))
    assert st == 0
    ref = oracle.filter_grid(box, 48.0, 0, 5.0)
    np.testing.assert_allclose(res, ref, atol=2e-5 * np.abs(ref).max())
    # undefined filter -> ValueError code 3 (reference Throw(ValueError), filtering.c:42-44)
    assert gpu_lib.test_filter(box.ctypes.data_as(C.c_void_p), 5.0, 0.0, 0.0, 7,
                               res.ctypes.data_as(C.c_void_p)) == 3


def test_in_loop_kernel_timing_hook(gpu_lib):
    """c21hip_ktime_enable / _report (bench.py's in-loop launch durations): events around every pass
    launch while enabled, summed per kernel kind; off by default and cleared on enable."""
    import ctypes as C

    import torch

    lib = gpu_lib
    lib.c21hip_ktime_report.argtypes = [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int)]
    lib.c21hip_bench_pass.restype = C.c_int
    lib.c21hip_bench_pass.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float,
                                      C.c_double, C.c_int, C.c_void_p, C.POINTER(C.c_float)]
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    ms = C.c_float()

    def count(kind):
        tot, cnt = C.c_double(), C.c_int()
        assert lib.c21hip_ktime_report(kind, C.byref(tot), C.byref(cnt)) == 0
        return cnt.value, tot.value

    lib.c21hip_ktime_enable(0)
    assert lib.c21hip_bench_pass(1, 128, 0, 3, 5.0, 37.0, 192.0, 3, stream, C.byref(ms)) == 0
    assert count(1)[0] == 0
    lib.c21hip_ktime_enable(1)
    assert lib.c21hip_bench_pass(1, 128, 0, 3, 5.0, 37.0, 192.0, 3, stream, C.byref(ms)) == 0
    n, total = count(1)
    assert n == 5 and total > 0  # two warm-up launches + three timed ones of pass Y
    assert total / n == pytest.approx(ms.value, rel=0.5)
    assert count(2)[0] == 0
    lib.c21hip_ktime_enable(0)
    assert count(1)[0] == 0
